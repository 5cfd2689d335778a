"""GPU: the drop-in lvmhelpers modules against golden fixtures produced by the REFERENCE's own
classes (tests/golden/make_goldens.py).  Index lists, supports and compaction order must be
identical; losses, log values and gradients within 2e-5 relative (the toy decoders' matmuls run
on the GPU here and on the CPU in the fixture; the kernels themselves are held to 1e-6 in
test_gpu_sparsemax.py / test_gpu_sparsemap.py)."""
import os

import numpy as np
import pytest
import torch

from toy_models import (bit_loss, cat_loss, flat_grads, load_flat_params, make_bit_problem,
                        make_cat_problem)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _close(a, b, rtol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    assert np.all(np.abs(a - b) <= rtol * scale), float(np.abs(a - b).max())


@pytest.mark.parametrize("name,normalizer", [("marg_softmax_cfg1", "softmax"),
                                             ("marg_sparsemax_cfg1", "sparsemax"),
                                             ("marg_sparsemax_cfg2", "sparsemax")])
def test_marginalizer_matches_reference(dev, name, normalizer):
    from lvmhelpers.marg import ExplicitWrapper, Marginalizer
    from lvmhelpers.utils import DeterministicWrapper
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, D, K, seed = (int(v) for v in g["meta"])
    enc, dec, x, labels = make_cat_problem(B, D, K, seed)
    load_flat_params(g["params"], enc, dec)
    assert np.array_equal(x.numpy(), g["x"]) and np.array_equal(labels.numpy(), g["labels"])
    enc, dec, x, labels = enc.to(dev), dec.to(dev), x.to(dev), labels.to(dev)
    model = Marginalizer(ExplicitWrapper(enc, normalizer=normalizer), DeterministicWrapper(dec),
                         cat_loss, encoder_entropy_coeff=float(g["coeff"]))
    res = model(x, x, labels)
    res["loss"].backward()
    lg = res["log"]
    assert sorted(lg.keys()) == list(g["log_keys"])
    assert res["loss"].dim() == 0 and lg["support"].dim() == 0 and lg["distr"].shape == (B, K)
    d = lg["distr"].detach().cpu().numpy()
    if normalizer == "sparsemax":
        assert np.array_equal(d > 0, g["log_distr"] > 0)          # exact support
    _close(d, g["log_distr"], 1e-5)
    assert float(lg["support"]) == float(g["log_support"])
    _close(res["loss"].item(), g["loss"])
    _close(lg["loss"].item(), g["log_loss"])
    _close(lg["acc"].item(), g["log_acc"])
    _close(lg["encoder_entropy"].item(), g["log_entropy"])
    _close(flat_grads(enc, dec), g["grads"])


def test_topk_sparsemax_marg_matches_reference(dev):
    from lvmhelpers.structmarg import TopKSparsemaxMarg, TopKSparsemaxWrapper
    from lvmhelpers.utils import DeterministicWrapper
    g = np.load(os.path.join(GOLD, "topk_marg.npz"))
    B, D, n, k, seed = (int(v) for v in g["meta"])
    enc, dec, x = make_bit_problem(B, D, n, seed)
    load_flat_params(g["params"], enc, dec)
    enc, dec, x = enc.to(dev), dec.to(dev), x.to(dev)
    model = TopKSparsemaxMarg(TopKSparsemaxWrapper(enc, k=k), DeterministicWrapper(dec), bit_loss,
                              encoder_entropy_coeff=1.0)
    res = model(x, x, x)
    res["loss"].backward()
    lg = res["log"]
    assert sorted(lg.keys()) == list(g["log_keys"])
    d = lg["distr"].detach().cpu().numpy()
    assert np.array_equal(d > 0, g["log_distr"] > 0)
    assert np.array_equal(lg["support"].cpu().numpy(), g["log_support"])
    assert lg["loss_output"].shape == g["log_loss_output"].shape       # same N, same order:
    _close(lg["loss_output"].detach().cpu().numpy(), g["log_loss_output"])
    _close(d, g["log_distr"], 1e-5)
    _close(res["loss"].item(), g["loss"])
    _close(lg["loss"].item(), g["log_loss"])
    _close(lg["encoder_entropy"].item(), g["log_entropy"])
    _close(flat_grads(enc, dec), g["grads"])


@pytest.mark.parametrize("name", ["sparsemap_marg", "sparsemap_budget_marg", "sparsemap_noinit_marg"])
def test_sparsemap_marg_matches_reference(dev, name):
    from lvmhelpers.structmarg import SparseMAPMarg, SparseMAPWrapper
    from lvmhelpers.utils import DeterministicWrapper
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, D, n, budget, init, seed = (int(v) for v in g["meta"])
    enc, dec, x = make_bit_problem(B, D, n, seed)
    load_flat_params(g["params"], enc, dec)
    enc, dec, x = enc.to(dev), dec.to(dev), x.to(dev)
    torch.manual_seed(42)
    model = SparseMAPMarg(SparseMAPWrapper(enc, budget=budget, init=bool(init)),
                          DeterministicWrapper(dec), bit_loss, encoder_entropy_coeff=1.0)
    res = model(x, x, x)
    res["loss"].backward()
    lg = res["log"]
    assert sorted(lg.keys()) == list(g["log_keys"])
    assert isinstance(lg["idxs"], list) and lg["idxs"] == g["log_idxs"].tolist()   # exact index list
    assert np.array_equal(lg["support"].cpu().numpy(), g["log_support"])
    _close(lg["distr"].cpu().numpy(), g["log_distr"], 1e-5)
    _close(lg["loss_output"].cpu().numpy(), g["log_loss_output"])
    _close(res["loss"].item(), g["loss"])
    _close(lg["encoder_entropy"].item(), g["log_entropy"])
    _close(flat_grads(enc, dec), g["grads"])


def test_per_sample_smap_api(dev, oracle_mod):
    from lvmhelpers.sparsemap import bernoulli_smap, budget_smap
    torch.manual_seed(42)
    x = torch.randn(5)
    xd = x.to(dev).requires_grad_(True)
    torch.manual_seed(1)
    p, aset = bernoulli_smap(xd)
    torch.manual_seed(1)
    init = torch.rand(10, dtype=torch.double).numpy()[None]
    ref = oracle_mod.sparsemap_fwd(x.numpy()[None], init, 0, 100)
    m = int(ref["alen"][0])
    assert p.shape == (m,) and aset.shape == (m, 5) and aset.dtype == torch.float32
    assert np.array_equal(p.detach().cpu().numpy(), ref["p"][0, :m])
    assert np.array_equal(aset.cpu().numpy(), oracle_mod.unpack_bits(ref["aset_bits"][0, :m], 5))
    (gx,) = torch.autograd.grad(p[0], xd)
    dp = np.zeros((1, 6), np.float32); dp[0, 0] = 1
    assert np.allclose(gx.cpu().numpy(), oracle_mod.sparsemap_bwd(dp, ref["aset_bits"], ref["alen"], ref["invA"], 5)[0], atol=1e-6)
    p2, aset2 = budget_smap(xd, budget=3)
    assert torch.all(aset2.sum(-1) <= 3) and abs(float(p2.detach().sum()) - 1) < 1e-6


def test_entmax_shim(dev, oracle_mod):
    import entmax
    rng = np.random.default_rng(0)
    x = rng.standard_normal((64, 10)).astype(np.float32)
    t = rng.integers(0, 10, 64)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    loss = entmax.SparsemaxLoss(reduction='none')(xt, torch.from_numpy(t).to(dev))
    loss.sum().backward()
    l_ref, g_ref = oracle_mod.sparsemax_loss(x, t)
    assert np.allclose(loss.detach().cpu().numpy(), l_ref, rtol=1e-5, atol=1e-6)
    assert np.allclose(xt.grad.cpu().numpy(), g_ref, atol=1e-6)
    p15 = entmax.entmax15(torch.from_numpy(x).to(dev), dim=-1)
    assert torch.allclose(p15.sum(-1), torch.ones(64, device=dev), atol=1e-5)


@pytest.mark.parametrize("K,coeff", [(10, 1.0), (256, 0.1)])
def test_marginalizer_compact_equals_dense(dev, K, coeff):
    """Support-only mode (one decoder call over the compacted (row, latent) pairs) gives the same
    loss, logs and gradients as the reference's column loop (marg.py:86-126)."""
    from lvmhelpers.marg import ExplicitWrapper, Marginalizer
    from lvmhelpers.utils import DeterministicWrapper
    from toy_models import CatDecoder, ScoreTable, cat_loss, _seeded
    B, D = 32, 12
    res = []
    for compact in (False, True):
        torch.manual_seed(0)
        enc = ScoreTable(B, K, scale=1.0)
        with torch.no_grad():
            enc.table.copy_(torch.randn(B, K, generator=torch.Generator().manual_seed(5)) * 1.5)
        dec = _seeded(CatDecoder(K, D), 11)
        enc, dec = enc.to(dev), dec.to(dev)
        x = torch.randn(B, D, generator=torch.Generator().manual_seed(6)).to(dev)
        labels = torch.randint(0, K, (B,), generator=torch.Generator().manual_seed(7)).to(dev)
        model = Marginalizer(ExplicitWrapper(enc, normalizer="sparsemax"), DeterministicWrapper(dec),
                             cat_loss, encoder_entropy_coeff=coeff, compact=compact)
        out = model(x, x, labels)
        out["loss"].backward()
        res.append((out, [p.grad.clone() for p in list(enc.parameters()) + list(dec.parameters())]))
    (o0, g0), (o1, g1) = res
    assert sorted(o0["log"].keys()) == sorted(o1["log"].keys())
    assert abs(float(o0["loss"]) - float(o1["loss"])) <= 2e-6 * max(1.0, abs(float(o0["loss"])))
    for k in ("loss", "acc", "encoder_entropy", "support"):
        assert abs(float(o0["log"][k]) - float(o1["log"][k])) <= 2e-6 * max(1.0, abs(float(o0["log"][k])))
    assert torch.equal(o0["log"]["distr"], o1["log"]["distr"])
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
