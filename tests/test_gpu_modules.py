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


@pytest.mark.parametrize("budget", [0, 3])
def test_pfactor_call_sequence(dev, oracle_mod, budget):
    """PFactorBernoulli / PFactorBudget driven exactly as the reference's run_sparsemap / jv drive
    them (lvmhelpers/sparsemap.py:15-30,42): initialize -> init_active_set_from_scores -> solve_qp
    -> get_sparse_solution -> dist_jacobian_vec, on host arrays, against the oracle."""
    from lvmhelpers.pbernoulli import PFactorBernoulli
    from lvmhelpers.pbudget import PFactorBudget
    n = 7
    x = torch.randn(n, generator=torch.Generator().manual_seed(9))
    f = PFactorBudget() if budget else PFactorBernoulli()
    f.initialize(n, budget) if budget else f.initialize(n)
    x_np = np.concatenate([x.numpy().astype(np.double), np.zeros(n)])            # sparsemap.py:21-22
    init = torch.rand(2 * n, dtype=torch.double, generator=torch.Generator().manual_seed(4))
    f.init_active_set_from_scores(init, [])                                      # sparsemap.py:27
    post, add = f.solve_qp(x_np, [], max_iter=100)                               # sparsemap.py:29
    aset, p = f.get_sparse_solution()                                            # sparsemap.py:30
    p = p[:len(aset)]
    ref = oracle_mod.sparsemap_fwd(x.numpy()[None], init.numpy()[None], budget, 100)
    m = int(ref["alen"][0])
    assert len(aset) == m and all(len(c) == n and set(c) <= {0, 1} for c in aset)
    assert np.array_equal(np.asarray(aset, np.float32), oracle_mod.unpack_bits(ref["aset_bits"][0, :m], n))
    assert np.array_equal(np.asarray(p, np.float32), ref["p"][0, :m])
    mu = (np.asarray(p)[:, None] * np.asarray(aset)).sum(0)
    assert post.shape == (2 * n,) and add.shape == (0,)
    np.testing.assert_allclose(post[:n], mu, atol=1e-7)
    np.testing.assert_allclose(post[:n] + post[n:], np.full(n, np.sum(p)), atol=1e-7)
    if budget == 0:                                                              # closed form
        np.testing.assert_allclose(post[:n], np.clip((x.numpy().astype(np.float64) + 1) / 2, 0, 1), atol=1e-6)
    else:
        assert all(sum(c) <= budget for c in aset)
    dp = np.random.default_rng(2).standard_normal(m).astype(np.float32)
    d_eta_u = np.empty(2 * n, dtype=np.float32)                                  # sparsemap.py:40-42
    d_eta_v = np.empty(0, dtype=np.float32)
    f.dist_jacobian_vec(dp, d_eta_u, d_eta_v)
    g = np.zeros((1, n + 1), np.float32)
    g[0, :m] = dp
    want = oracle_mod.sparsemap_bwd(g, ref["aset_bits"], ref["alen"], ref["invA"], n)[0]
    np.testing.assert_allclose(d_eta_u[:n], want, atol=1e-6)
    # potentials on both halves: only their difference matters; an unrepresentable one is refused
    f2 = PFactorBudget() if budget else PFactorBernoulli()
    f2.initialize(n, budget) if budget else f2.initialize(n)
    f2.init_active_set_from_scores(init, [])
    post2, _ = f2.solve_qp(np.concatenate([x_np[:n] + 0.5, np.full(n, 0.5)]), [], max_iter=100)
    np.testing.assert_allclose(post2, post, atol=1e-6)
    with pytest.raises(NotImplementedError):
        f2.solve_qp(np.concatenate([np.full(n, 1.0 / 3.0), np.zeros(n)]), [], max_iter=10)
    with pytest.raises(ValueError):
        f2.solve_qp(np.zeros(2 * n + 1), [], max_iter=10)


def test_autograd_class_names(dev, oracle_mod):
    """SparseMAP / BernSparseMAP / BudgetSparseMAP are importable and callable the way the
    reference's entry points call them (sparsemap.py:151-156), factor kept on ctx."""
    from lvmhelpers import sparsemap as sm
    assert issubclass(sm.BernSparseMAP, sm.SparseMAP) and issubclass(sm.BudgetSparseMAP, sm.SparseMAP)
    x = torch.randn(6, generator=torch.Generator().manual_seed(3)).to(dev).requires_grad_(True)
    torch.manual_seed(5)
    p, aset = sm.BernSparseMAP.apply(x, 100, True)
    torch.manual_seed(5)
    p_b, aset_b = sm.bernoulli_smap(x)
    assert torch.equal(p, p_b) and torch.equal(aset, aset_b) and not aset.requires_grad
    w = torch.randn(p.shape[0], generator=torch.Generator().manual_seed(1)).to(dev)
    (gx,) = torch.autograd.grad((p * w).sum(), x)
    torch.manual_seed(5)
    init = torch.rand(12, dtype=torch.double).numpy()[None]
    ref = oracle_mod.sparsemap_fwd(x.detach().cpu().numpy()[None], init, 0, 100)
    g = np.zeros((1, 7), np.float32)
    g[0, :p.shape[0]] = w.cpu().numpy()
    want = oracle_mod.sparsemap_bwd(g, ref["aset_bits"], ref["alen"], ref["invA"], 6)[0]
    np.testing.assert_allclose(gx.cpu().numpy(), want, atol=1e-6)
    p2, aset2 = sm.BudgetSparseMAP.apply(x, 2, 100, False)
    assert torch.all(aset2.sum(-1) <= 2) and abs(float(p2.detach().sum()) - 1) < 1e-6


def test_sequence_smap_as_the_reference_drives_it(dev, oracle_mod):
    """sequence_smap / SequenceBinarySparseMAP (sparsemap.py:47-87,130-148,159-160): p, aset and both
    gradients against the oracle with the same init draw; then PFactorSequenceBinary through the
    reference's own call sequence (initialize, set_additional_log_potentials,
    init_active_set_from_scores(init, zeros), solve_qp(eta_u, eta_v), get_sparse_solution,
    dist_jacobian_vec(dp, d_eta_u [2n], d_eta_v [n-1]))."""
    from lvmhelpers import sparsemap as sm
    from lvmhelpers.psequence import PFactorSequenceBinary
    assert issubclass(sm.SequenceBinarySparseMAP, sm.SequenceSparseMAP)
    n = 9
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, generator=g).to(dev).requires_grad_(True)
    t = torch.randn(n - 1, generator=g).to(dev).requires_grad_(True)
    torch.manual_seed(8)
    p, aset = sm.sequence_smap(x, t)
    assert not aset.requires_grad and aset.shape == (p.shape[0], n) and abs(float(p.detach().sum()) - 1) < 1e-6
    w = torch.randn(p.shape[0], generator=g).to(dev)
    gx, gt = torch.autograd.grad((p * w).sum(), (x, t))
    torch.manual_seed(8)
    init = torch.rand(2 * n, dtype=torch.double).numpy()[None]
    xn, tn = x.detach().cpu().numpy()[None], t.detach().cpu().numpy()[None]
    ref = oracle_mod.sparsemap_seq_fwd(xn, tn, init, max_iter=100)
    m = int(ref["alen"][0])
    assert p.shape[0] == m and np.array_equal(p.detach().cpu().numpy(), ref["p"][0, :m])
    assert np.array_equal(aset.cpu().numpy(), oracle_mod.unpack_bits(ref["aset_bits"], n)[0, :m])
    dp = np.zeros((1, n + 1), np.float32)
    dp[0, :m] = w.cpu().numpy()
    ds, dt = oracle_mod.sparsemap_seq_bwd(dp, ref["aset_bits"], ref["alen"], ref["invA"], n)
    np.testing.assert_allclose(gx.cpu().numpy(), ds[0], atol=1e-6)
    np.testing.assert_allclose(gt.cpu().numpy(), dt[0], atol=1e-6)
    # the factor object, as sparsemap.py:56-72,82-84 uses it
    f = PFactorSequenceBinary()
    f.initialize(n)
    x_np = np.concatenate([xn[0].astype(np.double), np.zeros(n)])
    t_np = tn[0].astype(np.double)
    f.set_additional_log_potentials(t_np)
    f.init_active_set_from_scores(torch.from_numpy(init[0]), torch.zeros(n - 1, dtype=torch.double))
    post, add = f.solve_qp(x_np, t_np, max_iter=100)
    aset_l, p_l = f.get_sparse_solution()
    assert len(aset_l) == m and np.allclose(p_l[:m], ref["p"][0, :m])
    A = np.asarray(aset_l, np.float64)
    np.testing.assert_allclose(post[:n], (np.asarray(p_l)[:, None] * A).sum(0), atol=1e-7)
    np.testing.assert_allclose(add, (np.asarray(p_l)[:, None] * A[:, :-1] * A[:, 1:]).sum(0), atol=1e-7)
    d_eta_u = np.empty(2 * n, dtype=np.float32)
    d_eta_v = np.empty(n - 1, dtype=np.float32)
    f.dist_jacobian_vec(w.cpu().numpy(), d_eta_u, d_eta_v)
    np.testing.assert_allclose(d_eta_u[:n], ds[0], atol=1e-6)
    np.testing.assert_allclose(d_eta_v, dt[0], atol=1e-6)
    with pytest.raises(ValueError):
        sm.sequence_smap(x, t[:-1])


def test_sparsemap_wrapper_is_world_size_independent(dev):
    """Data parallel: a batch of 128 solved whole, and as two 64-row shards whose init scores are
    the rank's rows of the GLOBAL draw (SparseMAPWrapper.shard -> dataparallel.global_init_scores):
    identical idxs / distr / support, and the row-weighted shard gradients equal the whole-batch
    gradient.  (The encoder is elementwise so the scores of a row do not depend on its batch.)"""
    from lvmhelpers.structmarg import SparseMAPMarg, SparseMAPWrapper
    from lvmhelpers.utils import DeterministicWrapper
    from toy_models import BitDecoder, _seeded
    B, n = 128, 32

    class Elem(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.full((n,), 1.5))
            self.b = torch.nn.Parameter(torch.zeros(n))

        def forward(self, x):
            return x * self.w + self.b

    x = torch.randn(B, n, generator=torch.Generator().manual_seed(21)).to(dev)

    def run(rows, shard):
        enc, dec = Elem().to(dev), _seeded(BitDecoder(n, n), 17).to(dev)
        wrap = SparseMAPWrapper(enc, budget=0, init=True)
        wrap.shard = shard
        model = SparseMAPMarg(wrap, DeterministicWrapper(dec), bit_loss, encoder_entropy_coeff=1.0)
        torch.manual_seed(33)                       # every rank starts from the same CPU stream
        out = model(x[rows], x[rows], x[rows])
        out["loss"].backward()
        grads = torch.cat([p.grad.reshape(-1) for p in list(enc.parameters()) + list(dec.parameters())])
        return out, grads

    whole, g_whole = run(slice(0, B), None)
    parts = [run(slice(r * 64, (r + 1) * 64), (r, 2, B)) for r in range(2)]
    idxs = [i + 64 * r for r, (o, _) in enumerate(parts) for i in o["log"]["idxs"]]
    assert idxs == whole["log"]["idxs"]
    assert torch.equal(torch.cat([o["log"]["distr"] for o, _ in parts]), whole["log"]["distr"])
    assert torch.equal(torch.cat([o["log"]["support"] for o, _ in parts]), whole["log"]["support"])
    g_parts = sum(64 / B * g for _, g in parts)
    assert float((g_parts - g_whole).abs().max()) <= 2e-5 * max(1.0, float(g_whole.abs().max()))
    # without the shard description the two shards would both draw rows 0..63 of the stream
    same, _ = run(slice(64, 128), None)
    assert same["log"]["idxs"] != parts[1][0]["log"]["idxs"] or not torch.equal(
        same["log"]["distr"], parts[1][0]["log"]["distr"])
    with pytest.raises(ValueError):
        run(slice(0, 60), (0, 2, B))


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


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_narrow_dtypes_are_widened_and_float64_refused(dev, dtype):
    """Autocast hands the normalisers half / bfloat16 logits: they are computed in fp32 (the
    kernels' arithmetic) and come back in the input's dtype, gradients included; float64 is
    refused instead of being narrowed without notice."""
    from lvmhelpers import functional as F_
    g = torch.Generator().manual_seed(4)
    x = torch.randn(33, 50, generator=g).to(dtype).to(dev)
    for fn in (F_.sparsemax, F_.entmax15):
        xn = x.clone().requires_grad_(True)
        xw = x.float().requires_grad_(True)
        pn, pw = fn(xn, dim=-1), fn(xw, dim=-1)
        assert pn.dtype == dtype and torch.equal(pn, pw.to(dtype))
        dy = torch.randn(33, 50, generator=g).to(dev)
        (pn.float() * dy).sum().backward()
        (pw.to(dtype).float() * dy).sum().backward()
        assert xn.grad.dtype == dtype and torch.equal(xn.grad, xw.grad.to(dtype))
        with pytest.raises(TypeError):
            fn(x.double(), dim=-1)


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


def test_compact_mode_refuses_a_list_input_on_batch_axis_0(dev):
    """A list/tuple decoder input is one entry per slot (axis 0); with the default batch_dims (0, 0)
    the support-only mode used to gather the FEATURE axis of every entry with row indices.  It now
    raises and names the setting the signal game needs."""
    import bench_models as bm
    from lvmhelpers.marg import Marginalizer
    model, (x, recv, labels) = bm.make_signal_game(dev, 64, compact=True, hoist=False)
    recv_list = [recv[i] for i in range(recv.shape[0])]
    model.batch_dims = (None, 0)
    with pytest.raises(ValueError, match="none,1"):
        model(x, recv_list, labels)
    # an axis that is not the batch axis is refused on the host (x is [game_size, B, feat]: axis 0
    # has 16 entries, the rows go up to 63 -- gathering it would read out of bounds on the GPU)
    model.batch_dims = (0, 1)
    with pytest.raises(ValueError, match="not the batch axis"):
        model(x, recv_list, labels)
    model.batch_dims = (None, 1)
    assert torch.isfinite(model(x, recv_list, labels)["loss"])


def _signal_game_run(dev, compact, hoist, list_input=False):
    import bench_models as bm
    model, (x, recv, labels) = bm.make_signal_game(dev, 64, compact=compact, hoist=hoist)
    if list_input:   # the Receiver only indexes x[i]: a list of game_size [B, feat] tensors works too
        recv = [recv[i] for i in range(recv.shape[0])]
    out = model(x, recv, labels)
    out["loss"].backward()
    return out, [p.grad.clone() for p in model.parameters()]


def test_receiver_hoisting_equals_dense_column_loop(dev):
    """SURVEY 8(f).2 on the config-2 shape (B=64, 16 images of 4096 features, vocabulary 256): the
    hoisted decoder path -- images embedded once per batch, (row, z) pairs gather the embeddings, one
    bmm (experiments/signal-game/archs.py:72-105 restructured) -- against the reference's column
    loop (lvmhelpers/marg.py:86-126) and against the plain support-only mode: same loss, logs and
    parameter gradients (fp32 sums in a different order: 1e-5 relative)."""
    o0, g0 = _signal_game_run(dev, False, False)
    o1, g1 = _signal_game_run(dev, True, False)
    o2, g2 = _signal_game_run(dev, True, True)
    o3, g3 = _signal_game_run(dev, True, True, list_input=True)
    assert float(o0["log"]["support"]) > 1.0          # a real marginalisation, not one column per row
    for o, g in ((o1, g1), (o2, g2), (o3, g3)):
        assert sorted(o0["log"].keys()) == sorted(o["log"].keys())
        assert abs(float(o0["loss"]) - float(o["loss"])) <= 1e-5 * max(1.0, abs(float(o0["loss"])))
        for k in ("loss", "acc", "encoder_entropy", "support"):
            assert abs(float(o0["log"][k]) - float(o["log"][k])) <= 1e-5 * max(1.0, abs(float(o0["log"][k])))
        assert torch.equal(o0["log"]["distr"], o["log"]["distr"])
        for a, b in zip(g0, g):
            assert float((a - b).abs().max()) <= 1e-5 * max(1e-3, float(a.abs().max()))


def test_hoisting_hook_protocol(dev):
    """A decoder takes part with embed_once/decode_pairs; hoist=True on a decoder without either
    interface is an error, hoist=False keeps the gathered call."""
    from lvmhelpers import fused
    from lvmhelpers.marg import ExplicitWrapper, Marginalizer
    from lvmhelpers.utils import DeterministicWrapper
    from toy_models import CatDecoder, ScoreTable, cat_loss, _seeded
    B, D, K = 16, 6, 12

    class TwoPhase(CatDecoder):
        calls = 0

        def embed_once(self, decoder_input):
            TwoPhase.calls += 1
            return decoder_input * 1.0

        def decode_pairs(self, z, state, rows):
            return self.forward(z, state.index_select(0, rows))

    outs = []
    for cls, hoist in ((CatDecoder, None), (TwoPhase, None), (TwoPhase, False)):
        enc = ScoreTable(B, K, scale=1.0)
        with torch.no_grad():
            enc.table.copy_(torch.randn(B, K, generator=torch.Generator().manual_seed(5)) * 1.5)
        dec = _seeded(cls(K, D), 11)
        x = torch.randn(B, D, generator=torch.Generator().manual_seed(6)).to(dev)
        labels = torch.randint(0, K, (B,), generator=torch.Generator().manual_seed(7)).to(dev)
        model = Marginalizer(ExplicitWrapper(enc.to(dev), normalizer="sparsemax"),
                             DeterministicWrapper(dec.to(dev)), cat_loss, compact=True, hoist=hoist)
        outs.append(float(model(x, x, labels)["loss"]))
    assert TwoPhase.calls == 1
    assert max(abs(outs[0] - outs[1]), abs(outs[1] - outs[2])) <= 1e-6 * max(1.0, abs(outs[0]))
    assert fused.hoisted_decoder(DeterministicWrapper(CatDecoder(K, D))) is None
    with pytest.raises(TypeError):
        Marginalizer(ExplicitWrapper(enc, normalizer="sparsemax"), DeterministicWrapper(CatDecoder(K, D).to(dev)),
                     cat_loss, compact=True, hoist=True)(x, x, labels)


def test_empty_batches(dev):
    """Zero rows / samples go through every wrapper of the path: empty outputs of the right shapes,
    no launch, no error (the last, ragged shard of a data-parallel split can be empty)."""
    from lvmhelpers import _native, functional as F_
    x = torch.empty(0, 17, device=dev, requires_grad=True)
    for fn in (F_.sparsemax, F_.entmax15):
        p = fn(x, dim=-1)
        assert tuple(p.shape) == (0, 17)
        p.sum().backward()
        assert tuple(x.grad.shape) == (0, 17)
    p = torch.empty(0, 17, device=dev)
    off, rows, cols, vals = _native.support_compact(p)
    assert off.tolist() == [0] and rows.numel() == cols.numel() == vals.numel() == 0
    cfg, top = _native.binary_topk_batched(torch.empty(0, 12, device=dev), 4, want_scores=True)
    assert tuple(cfg.shape) == (0, 4, 12) and tuple(top.shape) == (0, 4)
    out = _native.sparsemap_fwd(torch.empty(0, 12, device=dev), None, 0, 10)
    assert tuple(out["p"].shape) == (0, 13) and out["alen"].numel() == 0
    loss, lse, nll = _native.lowrank_ce_fwd(torch.empty(0, 5, 3, device=dev), torch.randn(7, 3, device=dev),
                                            torch.randn(7, device=dev), torch.empty(0, 5, dtype=torch.uint8, device=dev))
    assert loss.numel() == 0 and tuple(lse.shape) == (0, 5)


def test_embed_pairs_gradient_is_the_embedding_gradient(dev):
    """fused.embed_pairs: the lookup of the hoisted Receiver's message embedding, whose backward is a
    one-hot GEMM instead of torch's embedding_dense_backward -- same values, same gradient (a sum in
    another order), repeated and unused symbols included; other kinds of embedding fall through."""
    from lvmhelpers import fused
    torch.manual_seed(0)
    emb = torch.nn.Embedding(37, 24).to(dev)
    ref = torch.nn.Embedding(37, 24).to(dev)
    ref.load_state_dict(emb.state_dict())
    idx = torch.randint(0, 30, (500,), device=dev)          # symbols 30..36 never used
    w = torch.randn(500, 24, device=dev)
    out = fused.embed_pairs(emb, idx)
    (out * w).sum().backward()
    (ref(idx) * w).sum().backward()
    assert torch.equal(out, ref(idx))
    assert float((emb.weight.grad - ref.weight.grad).abs().max()) <= 1e-5 * float(ref.weight.grad.abs().max())
    assert float(emb.weight.grad[30:].abs().max()) == 0.0
    padded = torch.nn.Embedding(37, 24, padding_idx=0).to(dev)
    assert fused.embed_pairs(padded, idx).grad_fn.name().startswith("Embedding")
