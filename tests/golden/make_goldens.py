#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the build container, where
/root/reference exists:  python tests/golden/make_goldens.py).

What is pinned by what:
  binary_topk.npz      inputs -> outputs of the COMPILED REFERENCE (oracle/_ref, built from the
                       reference's own lvmhelpers/binary_topk.h): a true reference golden.
  binary_topk_wide.npz the same for k = 65 ... 1024 (configurations as packed bits).
  marg_*.npz           the reference's own Marginalizer / ExplicitWrapper classes
                       (lvmhelpers/marg.py, imported from /root/reference as text-unmodified
                       modules) run on CPU with toy encoder/decoder/loss.  With the softmax
                       normaliser nothing is stubbed in the arithmetic: pure reference output.
                       With sparsemax, the missing third-party `entmax` is replaced by the CPU
                       oracle, so those files pin the reference's LOOP semantics (log keys, which
                       sample the loss sees, reductions, gradients), not the sparsemax arithmetic
                       (parity unpinned, see oracle/sparsemax_oracle.c).
  topk_marg.npz        the reference's TopKSparsemaxWrapper/-Marg (lvmhelpers/structmarg.py) with
                       `.pbinary_topk` = the compiled reference and `entmax` = the oracle.
  sparsemap_marg.npz   the reference's SparseMAPWrapper/-Marg with `.sparsemap` = the oracle's
                       active-set solver (lpsmap is absent: parity unpinned for the solver).
Only data is written: inputs, parameters, expected outputs.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference/lvmhelpers"


# ------------------------------------------------------------------ stand-ins for absent packages
class _OracleSparsemax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        p, s, _, _ = oracle.sparsemax_fwd(X.detach().numpy().reshape(-1, X.shape[-1]))
        ctx.p, ctx.s = p, s
        return torch.from_numpy(p).view(X.shape)

    @staticmethod
    def backward(ctx, dY):
        dx = oracle.sparsemax_bwd(dY.contiguous().numpy().reshape(ctx.p.shape), ctx.p, ctx.s)
        return torch.from_numpy(dx).view(dY.shape)


def _sparsemax(X, dim=-1):
    assert dim in (-1, X.dim() - 1)
    return _OracleSparsemax.apply(X)


class _OracleSmap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, budget, max_iter, init):
        n = x.shape[0]
        init_scores = torch.rand(2 * n, dtype=torch.double).numpy()[None] if init else None
        o = oracle.sparsemap_fwd(x.detach().numpy()[None], init_scores, budget, max_iter)
        m = int(o["alen"][0])
        ctx.o, ctx.n, ctx.m = o, n, m
        aset = torch.from_numpy(oracle.unpack_bits(o["aset_bits"][0, :m], n))
        ctx.mark_non_differentiable(aset)
        return torch.from_numpy(o["p"][0, :m].copy()), aset

    @staticmethod
    def backward(ctx, dp, daset):
        g = np.zeros((1, ctx.n + 1), np.float32)
        g[0, :ctx.m] = dp.numpy()
        d = oracle.sparsemap_bwd(g, ctx.o["aset_bits"], ctx.o["alen"], ctx.o["invA"], ctx.n)
        return torch.from_numpy(d[0]), None, None, None


def _load_reference():
    """Import the reference's marg.py / structmarg.py under the package name `ref_lvmhelpers`."""
    ent = types.ModuleType("entmax")
    ent.sparsemax = _sparsemax
    ent.entmax15 = None
    sys.modules["entmax"] = ent
    pkg = types.ModuleType("ref_lvmhelpers")
    pkg.__path__ = [REF]
    sys.modules["ref_lvmhelpers"] = pkg
    smap = types.ModuleType("ref_lvmhelpers.sparsemap")
    smap.bernoulli_smap = lambda x, max_iter=100, init=True: _OracleSmap.apply(x, 0, max_iter, init)
    smap.budget_smap = lambda x, budget=5, max_iter=100, init=True: _OracleSmap.apply(x, budget, max_iter, init)
    sys.modules["ref_lvmhelpers.sparsemap"] = smap
    topk = types.ModuleType("ref_lvmhelpers.pbinary_topk")

    def batched_topk(scores, configs, k):
        configs[...] = oracle.ref_binary_topk_batched(np.asarray(scores), k)[0]
    topk.batched_topk = batched_topk
    sys.modules["ref_lvmhelpers.pbinary_topk"] = topk
    mods = {}
    for name in ("marg", "structmarg", "utils"):
        spec = importlib.util.spec_from_file_location(f"ref_lvmhelpers.{name}", os.path.join(REF, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


# ------------------------------------------------------------------ toy networks (shared with tests)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from toy_models import (CatDecoder, BitDecoder, cat_loss, bit_loss, make_cat_problem,  # noqa: E402
                        make_bit_problem, flat_params, flat_grads)


def gold_binary_topk():
    rng = np.random.default_rng(20201206)
    out = {}
    cases = [(64, 32, 10), (64, 128, 10), (16, 4, 16), (8, 1, 2), (8, 2, 4), (8, 33, 2),
             (8, 64, 17), (4, 512, 64), (32, 20, 64), (4, 5, 32)]
    for i, (B, n, k) in enumerate(cases):
        x = rng.standard_normal((B, n)).astype(np.float32)
        x[::3] = np.round(x[::3] * 2) / 2
        if B > 2:
            x[1] = 0.0
        cfg, top = oracle.ref_binary_topk_batched(x, k)
        out[f"scores_{i}"], out[f"k_{i}"], out[f"configs_{i}"], out[f"top_{i}"] = x, np.int32(k), cfg.astype(np.uint8), top
    # the probe recorded in SURVEY.md 8(c)
    cfg, top = oracle.ref_binary_topk_batched(np.array([[0.5, -1, 2, 0.1]], np.float32), 5)
    out["probe_configs"], out["probe_top"] = cfg.astype(np.uint8), top
    np.savez_compressed(os.path.join(HERE, "binary_topk.npz"), n_cases=len(cases), **out)


def gold_binary_topk_wide():
    """k > 64 (round 3: the workgroup-per-sample kernel), same compiled reference, a file of its own so
    that binary_topk.npz stays byte for byte what round 1 wrote."""
    rng = np.random.default_rng(20261111)
    out = {}
    cases = [(6, 32, 65), (5, 7, 128), (4, 100, 100), (3, 40, 256), (3, 33, 257), (2, 512, 300),
             (2, 64, 1000), (2, 200, 1024), (1, 10, 1024)]
    for i, (B, n, k) in enumerate(cases):
        x = rng.standard_normal((B, n)).astype(np.float32)
        x[::3] = np.round(x[::3] * 2) / 2
        if B > 2:
            x[1] = 0.0
        cfg, top = oracle.ref_binary_topk_batched(x, k)
        # configurations as packed bits (k * n up to 204800 per sample)
        out[f"scores_{i}"], out[f"k_{i}"], out[f"top_{i}"] = x, np.int32(k), top
        out[f"configs_packed_{i}"] = np.packbits(cfg.astype(np.uint8), axis=-1)
    np.savez_compressed(os.path.join(HERE, "binary_topk_wide.npz"), n_cases=len(cases), **out)


def _run_marg(ref, normalizer, B, D, K, coeff, seed):
    torch.manual_seed(seed)
    enc, dec, x, labels = make_cat_problem(B, D, K, seed)
    model = ref["marg"].Marginalizer(
        ref["marg"].ExplicitWrapper(enc, normalizer=normalizer),
        ref["utils"].DeterministicWrapper(dec), cat_loss, encoder_entropy_coeff=coeff)
    res = model(x, x, labels)
    res["loss"].backward()
    return dict(
        x=x.numpy(), labels=labels.numpy(), params=flat_params(enc, dec), grads=flat_grads(enc, dec),
        loss=res["loss"].detach().numpy(), log_loss=res["log"]["loss"].detach().numpy(),
        log_acc=res["log"]["acc"].detach().numpy(),
        log_entropy=res["log"]["encoder_entropy"].detach().numpy(),
        log_support=res["log"]["support"].numpy(), log_distr=res["log"]["distr"].detach().numpy(),
        log_keys=np.array(sorted(res["log"].keys())), meta=np.array([B, D, K, seed]), coeff=np.float64(coeff))


def gold_marg(ref):
    for name, normalizer, B, D, K, coeff in (
            ("marg_softmax_cfg1", "softmax", 64, 20, 10, 1.0),
            ("marg_sparsemax_cfg1", "sparsemax", 64, 20, 10, 1.0),
            ("marg_sparsemax_cfg2", "sparsemax", 64, 24, 256, 0.1)):
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **_run_marg(ref, normalizer, B, D, K, coeff, 7))


def gold_topk_marg(ref):
    B, D, n, k = 64, 24, 32, 10
    enc, dec, x = make_bit_problem(B, D, n, 11)
    model = ref["structmarg"].TopKSparsemaxMarg(
        ref["structmarg"].TopKSparsemaxWrapper(enc, k=k), ref["utils"].DeterministicWrapper(dec),
        bit_loss, encoder_entropy_coeff=1.0)
    res = model(x, x, x)
    res["loss"].backward()
    lg = res["log"]
    np.savez_compressed(
        os.path.join(HERE, "topk_marg.npz"), x=x.numpy(), params=flat_params(enc, dec),
        grads=flat_grads(enc, dec), loss=res["loss"].detach().numpy(), log_loss=lg["loss"].numpy(),
        log_entropy=lg["encoder_entropy"].numpy(), log_support=lg["support"].numpy(),
        log_distr=lg["distr"].detach().numpy(), log_loss_output=lg["loss_output"].detach().numpy(),
        log_keys=np.array(sorted(lg.keys())), meta=np.array([B, D, n, k, 11]))


def gold_sparsemap_marg(ref):
    for name, n, budget, init in (("sparsemap_marg", 32, 0, True), ("sparsemap_budget_marg", 32, 16, True),
                                  ("sparsemap_noinit_marg", 32, 0, False)):
        B, D = 64, 24
        enc, dec, x = make_bit_problem(B, D, n, 13)
        torch.manual_seed(42)  # the init draws come from the global CPU generator (sparsemap.py:26)
        model = ref["structmarg"].SparseMAPMarg(
            ref["structmarg"].SparseMAPWrapper(enc, budget=budget, init=init),
            ref["utils"].DeterministicWrapper(dec), bit_loss, encoder_entropy_coeff=1.0)
        res = model(x, x, x)
        res["loss"].backward()
        lg = res["log"]
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), x=x.numpy(), params=flat_params(enc, dec),
            grads=flat_grads(enc, dec), loss=res["loss"].detach().numpy(), log_loss=lg["loss"].numpy(),
            log_entropy=lg["encoder_entropy"].numpy(), log_support=lg["support"].numpy(),
            log_distr=lg["distr"].numpy(), log_loss_output=lg["loss_output"].numpy(),
            log_idxs=np.array(lg["idxs"], np.int64), log_keys=np.array(sorted(lg.keys())),
            meta=np.array([B, D, n, budget, int(init), 13]))


def gold_common_params(ref):
    """The reference's own populate_common_params (lvmhelpers/utils.py:20-99) applied to a fresh
    parser: every flag with its type, default, choices and action kind, the namespace of an empty
    command line and of the command lines the published scripts use."""
    import argparse
    import json
    parser = argparse.ArgumentParser()
    ref["utils"].populate_common_params(parser)
    flags = []
    for a in parser._actions:
        if not a.option_strings or a.dest == "help":
            continue
        flags.append(dict(flags=a.option_strings, dest=a.dest, kind=type(a).__name__,
                          type=None if a.type is None else a.type.__name__, default=a.default,
                          choices=None if a.choices is None else list(a.choices), nargs=a.nargs))
    lines = [[], ["--mode", "marg", "--normalizer", "sparsemax", "--latent_size", "256", "--entropy_coeff", "0.1"],
             ["--mode", "sparsemap", "--budget", "16", "--noinit", "--batch_size", "64", "--latent_size", "32"],
             ["--mode", "topksparse", "--topksparse", "10", "--lr", "0.002", "--no_cuda"],
             ["--mode", "gs", "--gs_tau", "0.5", "--straight_through", "--random_seed", "7"]]
    parsed = [dict(argv=argv, namespace=vars(parser.parse_args(argv))) for argv in lines]
    with open(os.path.join(HERE, "common_params.json"), "w") as f:
        json.dump(dict(flags=flags, parsed=parsed), f, indent=1, sort_keys=True)


if __name__ == "__main__":
    assert os.path.isdir(REF), "needs /root/reference"
    oracle.build()
    only = sys.argv[1:]          # e.g. `make_goldens.py common_params`: regenerate just that fixture
    if only == ["binary_topk_wide"]:   # needs the compiled header only, none of the Python modules
        gold_binary_topk_wide()
        print("goldens written to", HERE)
        sys.exit(0)
    ref = _load_reference()
    if not only or "binary_topk" in only:
        gold_binary_topk()
    if not only or "binary_topk_wide" in only:
        gold_binary_topk_wide()
    if not only or "marg" in only:
        gold_marg(ref)
    if not only or "topk_marg" in only:
        gold_topk_marg(ref)
    if not only or "sparsemap_marg" in only:
        gold_sparsemap_marg(ref)
    if not only or "common_params" in only:
        gold_common_params(ref)
    print("goldens written to", HERE)
