"""GPU parity of kernel 2 (support compaction, offsets, weighted reduction) and the k-best
enumerator against the CPU oracle (bit-exact: integer/index work and fp32 index-order sums)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,K,density", [(64, 10, 0.3), (64, 256, 0.02), (1, 1, 1.0), (5, 3, 0.0),
                                         (100, 129, 0.5), (7, 4096, 0.001), (3, 65536, 0.3),
                                         (5000, 40, 0.1), (9, 1027, 1.0), (1001, 256, 0.05), (13, 252, 0.5),
                                         (4, 4, 1.0), (1 << 16, 256, 0.02)])
def test_compact_matches_oracle(dev, oracle_mod, R, K, density):
    from lvmhelpers import _native
    rng = np.random.default_rng(R * 131 + K)
    p = rng.random((R, K)).astype(np.float32)
    p[rng.random((R, K)) >= density] = 0.0
    off_ref, rows_ref, cols_ref, vals_ref = oracle_mod.support_compact(p)
    pt = torch.from_numpy(p).to(dev)
    nnz = _native.support_count(pt)
    assert np.array_equal(nnz.cpu().numpy(), np.diff(off_ref).astype(np.int32))
    off, rows, cols, vals = _native.support_compact(pt)
    assert np.array_equal(off.cpu().numpy(), off_ref)
    assert np.array_equal(rows.cpu().numpy(), rows_ref)
    assert np.array_equal(cols.cpu().numpy(), cols_ref)
    assert np.array_equal(vals.cpu().numpy(), vals_ref)
    # same order as torch's own boolean-mask / nonzero (lvmhelpers/structmarg.py:116-126)
    nz = torch.nonzero(pt > 0)
    assert torch.equal(nz[:, 0].to(torch.int32), rows) and torch.equal(nz[:, 1].to(torch.int32), cols)


def test_offsets_large(dev):
    from lvmhelpers import _native
    g = torch.Generator(device=dev).manual_seed(5)
    for R in (1, 255, 256, 2048, 2049, 1 << 20, (1 << 20) + 17):
        nnz = torch.randint(0, 9, (R,), device=dev, generator=g, dtype=torch.int32)
        off = _native.support_offsets(nnz)
        ref = torch.zeros(R + 1, dtype=torch.int64, device=dev)
        ref[1:] = torch.cumsum(nnz.to(torch.int64), 0)
        assert torch.equal(off, ref)


@pytest.mark.parametrize("N,R", [(0, 4), (5, 3), (640, 64), (100000, 777), (3000001, 1000)])
def test_weighted_loss_reduce(dev, oracle_mod, N, R):
    from lvmhelpers import _native, functional as F_
    rng = np.random.default_rng(N + R)
    vals = rng.random(N).astype(np.float32)
    comps = (500 * rng.random(N)).astype(np.float32)
    cuts = np.sort(rng.integers(0, N + 1, R - 1)) if R > 1 else np.zeros(0, np.int64)
    off = np.concatenate([[0], cuts, [N]]).astype(np.int64)
    out_ref, tot_ref = oracle_mod.weighted_loss_reduce(vals, comps, off, 1.0 / 64)
    vt, ct = torch.from_numpy(vals).to(dev), torch.from_numpy(comps).to(dev)
    total, rows = _native.weighted_loss_reduce_fwd(vt, ct, torch.from_numpy(off).to(dev), 1.0 / 64,
                                                   want_rows=True)
    np.testing.assert_allclose(total.item(), tot_ref, rtol=1e-6)
    np.testing.assert_allclose(rows.cpu().numpy(), out_ref, rtol=1e-6, atol=1e-30)
    if N:
        vt.requires_grad_(True); ct.requires_grad_(True)
        (3.0 * F_.weighted_loss(vt, ct, 1.0 / 64)).backward()
        np.testing.assert_allclose(vt.grad.cpu().numpy(), 3.0 / 64 * comps, rtol=1e-6)
        np.testing.assert_allclose(ct.grad.cpu().numpy(), 3.0 / 64 * vals, rtol=1e-6)


@pytest.mark.parametrize("B,n,k", [(64, 32, 10), (64, 128, 10), (8, 4, 16), (5, 1, 2), (3, 2, 4),
                                   (16, 33, 2), (7, 64, 17), (4, 512, 64), (300, 20, 64),
                                   (2, 5, 32), (1000, 128, 10)])
def test_binary_topk_matches_oracle(dev, oracle_mod, B, n, k):
    from lvmhelpers import _native
    rng = np.random.default_rng(B + 17 * n + k)
    scores = rng.standard_normal((B, n)).astype(np.float32)
    scores[::3] = np.round(scores[::3] * 2) / 2          # ties
    if B > 2:
        scores[1] = 0.0                                   # all zero: every configuration ties
    cfg_ref, top_ref = oracle_mod.binary_topk_batched(scores, k)
    cfg, top = _native.binary_topk_batched(torch.from_numpy(scores).to(dev), k, want_scores=True)
    assert np.array_equal(top.cpu().numpy(), top_ref)
    assert np.array_equal(cfg.cpu().numpy(), cfg_ref)


@pytest.mark.parametrize("B,n,k", [(5, 32, 65), (9, 7, 128), (4, 100, 100), (3, 40, 256), (3, 33, 257),
                                   (2, 512, 300), (2, 64, 1000), (2, 200, 1024), (1, 10, 1024), (70, 24, 90)])
def test_binary_topk_beyond_one_wave(dev, oracle_mod, B, n, k):
    """k > 64 (the reference takes any k, binary_topk.h:57-129): one workgroup per sample, merge
    ranks by binary search instead of a linear count -- configurations and fp32 scores bit for bit
    against the restatement and, where it was built, the reference's own header compiled in place."""
    from lvmhelpers import _native
    rng = np.random.default_rng(B + 17 * n + k)
    scores = rng.standard_normal((B, n)).astype(np.float32)
    scores[::3] = np.round(scores[::3] * 2) / 2          # ties
    if B > 2:
        scores[1] = 0.0                                   # all zero: every configuration ties
    cfg_ref, top_ref = oracle_mod.binary_topk_batched(scores, k)
    if oracle_mod.ref_lib() is not None:
        cfg_r, top_r = oracle_mod.ref_binary_topk_batched(scores, k)
        assert np.array_equal(cfg_r, cfg_ref) and np.array_equal(top_r, top_ref)
    cfg, top = _native.binary_topk_batched(torch.from_numpy(scores).to(dev), k, want_scores=True)
    assert np.array_equal(top.cpu().numpy(), top_ref)
    assert np.array_equal(cfg.cpu().numpy(), cfg_ref)


def test_binary_topk_rejects_undefined_cases(dev):
    from lvmhelpers import _native
    x = torch.zeros(2, 3, device=dev)
    with pytest.raises(_native.NativeError):
        _native.binary_topk_batched(x, 9)      # k > 2^n: UB in the reference
    with pytest.raises(_native.NativeError):
        _native.binary_topk_batched(x, 1)      # k < 2: UB in the reference
    with pytest.raises(_native.NativeError):
        _native.binary_topk_batched(torch.zeros(2, 40, device=dev), 1025)


def test_pbinary_topk_host_api(dev, oracle_mod):
    from lvmhelpers import pbinary_topk
    out = pbinary_topk.binary_topk([0.5, -1, 2, 0.1], 5)
    assert [c for _, c in out] == [[1, 0, 1, 1], [1, 0, 1, 0], [0, 0, 1, 1], [0, 0, 1, 0], [1, 1, 1, 1]]
    rng = np.random.default_rng(0)
    scores = rng.standard_normal((64, 32)).astype(np.float32)
    cfg = np.empty((64, 10, 32), np.float32)
    pbinary_topk.batched_topk(scores, cfg, 10)
    assert np.array_equal(cfg, oracle_mod.binary_topk_batched(scores, 10)[0])
    with pytest.raises(ValueError):
        pbinary_topk.batched_topk(scores.astype(np.float64), cfg, 10)


def test_binary_topk_kernel_on_the_reference_golden_vectors(dev):
    """tests/golden/binary_topk.npz was produced by the reference's own binary_topk.h compiled in
    place (oracle/_ref, tests/golden/make_goldens.py); every case that the kernel's limits admit
    (k <= 1024, n <= 512) goes straight to smlvm_binary_topk_batched: configurations and fp32 scores
    bit for bit (lvmhelpers/binary_topk.h:57-129, pbinary_topk.pyx:23-34)."""
    import os
    from lvmhelpers import _native
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "binary_topk.npz"))
    n_cases = int(d["n_cases"])
    assert n_cases >= 10
    for i in range(n_cases):
        scores, k = d[f"scores_{i}"], int(d[f"k_{i}"])
        cfg, top = _native.binary_topk_batched(torch.from_numpy(scores).to(dev), k, want_scores=True)
        assert np.array_equal(cfg.cpu().numpy().astype(np.uint8), d[f"configs_{i}"]), i
        assert np.array_equal(top.cpu().numpy(), d[f"top_{i}"]), i
    w = np.load(os.path.join(os.path.dirname(__file__), "golden", "binary_topk_wide.npz"))   # k = 65 ... 1024
    for i in range(int(w["n_cases"])):
        scores, k = w[f"scores_{i}"], int(w[f"k_{i}"])
        cfg, top = _native.binary_topk_batched(torch.from_numpy(scores).to(dev), k, want_scores=True)
        want = np.unpackbits(w[f"configs_packed_{i}"], axis=-1)[..., :scores.shape[1]]
        assert np.array_equal(cfg.cpu().numpy().astype(np.uint8), want), (i, k)
        assert np.array_equal(top.cpu().numpy(), w[f"top_{i}"]), (i, k)
    # SURVEY.md 8(c)'s probe: binary_topk([0.5, -1, 2, 0.1], 5)
    cfg, top = _native.binary_topk_batched(torch.tensor([[0.5, -1.0, 2.0, 0.1]], device=dev), 5, want_scores=True)
    assert np.array_equal(cfg.cpu().numpy().astype(np.uint8), d["probe_configs"])
    assert np.array_equal(top.cpu().numpy(), d["probe_top"])


@pytest.mark.parametrize("R,K,sigma", [(64, 10, 1.0), (64, 256, 1.0), (300, 1000, 1.0), (64, 4096, 1.0),
                                       (64, 4096, 0.05), (20, 8192, 1.0), (9, 65536, 1.0),
                                       (128, 256, 0.05)])
def test_compact_from_support_lists(dev, R, K, sigma):
    """The compaction fed with the forward's support lists (rows with a list never read their K
    probabilities; rows marked -2 are scanned) must equal the scan of p: same offsets, and rows /
    cols / vals in the row-major order of torch.nonzero(p > 0)."""
    from lvmhelpers import _native, functional as F_
    g = torch.Generator(device=dev).manual_seed(R + K)
    x = torch.empty(R, K, device=dev).normal_(0.0, sigma, generator=g)
    p, _, nnz, _, _, lists = _native.sparsemax_fwd(x, want_nnz=True, want_cols=True)
    off0, rows0, cols0, vals0 = _native.support_compact(p)
    off1, rows1, cols1, vals1 = _native.support_compact(p, support_cols=lists)
    assert torch.equal(off0, off1) and torch.equal(rows0, rows1)
    assert torch.equal(cols0, cols1) and torch.equal(vals0, vals1)
    nz = torch.nonzero(p > 0)
    assert torch.equal(rows1.long(), nz[:, 0]) and torch.equal(cols1.long(), nz[:, 1])
    assert torch.equal((off1[1:] - off1[:-1]).to(torch.int32), nnz)
    # the autograd wrappers tag their output with the lists
    q = F_.sparsemax(x.clone().requires_grad_(True))
    tagged = F_.support_cols_of(q)
    norm = lambda t: torch.where(t == -3, torch.full_like(t, -2), t)   # -3: overflow row with a support mask
    assert tagged is not None and torch.equal(norm(tagged), lists)
    assert F_.support_cols_of(q.detach()) is None and F_.support_cols_of(q * 1.0) is None
    q2, _, _ = F_.sparsemax_with_entropy(x)
    assert torch.equal(norm(F_.support_cols_of(q2)), lists)
    # compaction from lists whose overflow rows carry the mask mark
    off2, rows2, cols2, vals2 = _native.support_compact(p, support_cols=tagged)
    assert torch.equal(off2, off0) and torch.equal(cols2, cols0) and torch.equal(vals2, vals0)


@pytest.mark.parametrize("R,K,cap", [(1, 8, 16), (7, 12, 16), (33, 256, 32), (1000, 256, 32), (65, 1000, 64),
                                     (129, 250, 32), (50, 4096, 64), (9, 100, 8), (40, 64, 48), (257, 37, 32)])
def test_list_fed_kernels_on_handmade_lists(dev, oracle_mod, R, K, cap):
    """The list-fed count/compaction on lists built by hand: entries in random order, packed at the
    front and padded with -1; every third row unlisted (-2 or -3 in slot 0, garbage behind it),
    rows with an empty support, rows that fill the list; R ragged against the rows a wave / a
    workgroup takes, caps on and off the multi-row path (16/32/64 vs 8/48), K on and off the 16-byte
    path.  Expected: the oracle's compaction of p (row-major torch.nonzero order)."""
    from lvmhelpers import _native
    rng = np.random.default_rng(R * 7 + K * 3 + cap)
    p = np.zeros((R, K), np.float32)
    lists = np.full((R, cap), -1, np.int32)
    for r in range(R):
        kind = r % 6
        n = int(rng.integers(0, min(cap, K) + 1)) if kind not in (1, 4) else (min(cap, K) if kind == 1 else 0)
        cols = rng.choice(K, n, replace=False)
        if r % 3 == 2:     # no list: any support size
            cols = rng.choice(K, int(rng.integers(1, K + 1)), replace=False)
            lists[r] = rng.integers(-5, K, cap)
            lists[r, 0] = -2 if r % 2 else -3
        else:
            lists[r, :n] = cols
        p[r, cols] = rng.random(len(cols)).astype(np.float32) + 0.01
    off_ref, rows_ref, cols_ref, vals_ref = oracle_mod.support_compact(p)
    pt, lt = torch.from_numpy(p).to(dev), torch.from_numpy(lists).to(dev)
    off, rows, cols, vals = _native.support_compact(pt, support_cols=lt)
    assert np.array_equal(off.cpu().numpy(), off_ref)
    assert np.array_equal(rows.cpu().numpy(), rows_ref)
    assert np.array_equal(cols.cpu().numpy(), cols_ref)
    assert np.array_equal(vals.cpu().numpy(), vals_ref)


@pytest.mark.parametrize("R,K,sigma", [(64, 10, 1.0), (64, 256, 1.0), (300, 1000, 1.0), (64, 4096, 1.0),
                                       (64, 4096, 0.05), (1000, 256, 0.05), (5000, 256, 1.0), (20, 8192, 1.0),
                                       (257, 4096, 0.2), (1, 64, 1.0), (255, 128, 1.0), (256, 128, 1.0),
                                       (70000, 64, 1.0), (64, 33, 1.0), (513, 63, 0.3), (1000, 1, 1.0),
                                       (3, 2, 1.0), (4100, 17, 0.05)])
def test_fused_single_pass_compaction(dev, R, K, sigma):
    """smlvm_support_compact_fused (count + look-back scan + compaction in one launch) against the
    separate launches: same offsets / rows / cols / vals bit for bit -- fed with nothing (scan of
    p), with the forward's column lists (probabilities gathered), with column + value lists (nothing
    gathered), and with an entry capacity below N (the entries beyond it are dropped, the offsets
    stay complete).  The value lists hold the bits of p at the listed columns."""
    from lvmhelpers import _native
    g = torch.Generator(device=dev).manual_seed(R + K)
    x = torch.empty(R, K, device=dev).normal_(0.0, sigma, generator=g)
    if K <= 4096:
        p, _, nnz, _, _, lists, vlists = _native.sparsemax_fwd(x, want_nnz=True, want_cols=True, want_vals=True)
        p2, _, _, _, _, lists2 = _native.sparsemax_fwd(x, want_cols=True)
        assert torch.equal(p, p2) and torch.equal(lists, lists2)      # the value lists change nothing else
        listed = lists[:, 0] >= -1
        ok = lists >= 0
        got = vlists[ok & listed[:, None]]
        want = p.gather(1, lists.clamp(min=0).long())[ok & listed[:, None]]
        assert torch.equal(got, want)
    else:
        p, _, nnz, _, _, lists = _native.sparsemax_fwd(x, want_nnz=True, want_cols=True)
        lists, vlists = None, None    # 128-slot lists: the backward's only
    off0, rows0, cols0, vals0 = _native.support_compact(p, fused=False)
    N = int(off0[R])
    variants = [(None, None)]
    if lists is not None:
        variants += [(lists, None), (lists, vlists)]
    for sc, sv in variants:
        out = _native.support_compact_fused(p, sc, sv)
        assert out is not None
        off1, rows1, cols1, vals1 = out
        assert torch.equal(off0, off1) and torch.equal(rows0, rows1)
        assert torch.equal(cols0, cols1) and torch.equal(vals0, vals1)
        # capacity below N: a prefix of the entries, complete offsets
        half = N // 2
        off2, rows2, cols2, vals2 = _native.support_compact_fused(p, sc, sv, n_cap=half)
        assert torch.equal(off2, off0)   # (the wrapper redid the entries from the offsets: N > n_cap)
        assert torch.equal(rows2, rows0) and torch.equal(cols2, cols0) and torch.equal(vals2, vals0)
    assert torch.equal((off0[1:] - off0[:-1]).to(torch.int32), nnz)


@pytest.mark.parametrize("R,K,cap", [(1, 8, 16), (7, 12, 16), (33, 256, 32), (1000, 256, 32), (65, 1000, 64),
                                     (129, 250, 32), (50, 4096, 64), (257, 37, 32), (3000, 300, 16)])
def test_fused_compaction_on_handmade_lists(dev, oracle_mod, R, K, cap):
    """The single-pass compaction on hand-made lists (random order, -1 padding, every third row
    unlisted with garbage behind the mark, empty and full lists, ragged R) with and without value
    lists, and through the raw C ABI with a capacity below N: dropped entries, complete offsets."""
    import ctypes
    from lvmhelpers import _native
    rng = np.random.default_rng(R * 7 + K * 3 + cap)
    p = np.zeros((R, K), np.float32)
    lists = np.full((R, cap), -1, np.int32)
    vl = rng.random((R, cap)).astype(np.float32) * 0 - 7.0       # garbage where no column is listed
    for r in range(R):
        kind = r % 6
        n = int(rng.integers(0, min(cap, K) + 1)) if kind not in (1, 4) else (min(cap, K) if kind == 1 else 0)
        cols = rng.choice(K, n, replace=False)
        if r % 3 == 2:
            cols = rng.choice(K, int(rng.integers(1, K + 1)), replace=False)
            lists[r] = rng.integers(-5, K, cap)
            lists[r, 0] = -2 if r % 2 else -3
        else:
            lists[r, :n] = cols
        p[r, cols] = rng.random(len(cols)).astype(np.float32) + 0.01
        if r % 3 != 2:
            vl[r, :n] = p[r, cols]
    off_ref, rows_ref, cols_ref, vals_ref = oracle_mod.support_compact(p)
    pt, lt, vt = torch.from_numpy(p).to(dev), torch.from_numpy(lists).to(dev), torch.from_numpy(vl).to(dev)
    for sv in (None, vt):
        off, rows, cols, vals = _native.support_compact_fused(pt, lt, sv, n_cap=R * K)
        assert np.array_equal(off.cpu().numpy(), off_ref)
        assert np.array_equal(rows.cpu().numpy(), rows_ref)
        assert np.array_equal(cols.cpu().numpy(), cols_ref)
        assert np.array_equal(vals.cpu().numpy(), vals_ref)
    # raw ABI, capacity below N
    L = _native.lib()
    N = int(off_ref[-1])
    ncap = max(N - 5, 0)
    off = torch.empty(R + 1, dtype=torch.int64, device=dev)
    rows = torch.full((ncap + 8,), -9, dtype=torch.int32, device=dev)
    cols = torch.full((ncap + 8,), -9, dtype=torch.int32, device=dev)
    vals = torch.full((ncap + 8,), -9.0, dtype=torch.float32, device=dev)
    nb = L.smlvm_support_compact_fused_workspace_bytes(ctypes.c_int64(R))
    ws = torch.full((nb // 8,), -1, dtype=torch.int64, device=dev)     # dirty: the call clears it
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = L.smlvm_support_compact_fused(P(pt), P(lt), P(vt), ctypes.c_int32(cap), P(off), P(rows), P(cols), P(vals),
                                       ctypes.c_int64(ncap), ctypes.c_int64(R), ctypes.c_int32(K), P(ws),
                                       ctypes.c_size_t(nb), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(off.cpu().numpy(), off_ref)
    assert ws[1].item() == 0 and ws[2].item() == N
    assert np.array_equal(rows[:ncap].cpu().numpy(), rows_ref[:ncap]) and bool((rows[ncap:] == -9).all())
    assert np.array_equal(cols[:ncap].cpu().numpy(), cols_ref[:ncap]) and bool((cols[ncap:] == -9).all())
    assert np.array_equal(vals[:ncap].cpu().numpy(), vals_ref[:ncap]) and bool((vals[ncap:] == -9).all())


def test_support_tag_goes_stale_when_p_is_modified(dev):
    """The support lists ride on the probability tensor as a Python attribute; an in-place change of
    p afterwards must invalidate them (compaction would otherwise list the old support)."""
    import torch
    from lvmhelpers import functional as F_, _native
    x = torch.randn(32, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    p = F_.sparsemax(x, dim=-1)
    assert F_.support_cols_of(p) is not None
    off0, rows0, cols0, _ = _native.support_compact(p, support_cols=F_.support_cols_of(p))
    off1, rows1, cols1, _ = _native.support_compact(p)
    assert torch.equal(rows0, rows1) and torch.equal(cols0, cols1) and torch.equal(off0, off1)
    p[:, :32] = 0.0                                   # masks half of the categories in place
    assert F_.support_cols_of(p) is None
    _, rows2, cols2, _ = _native.support_compact(p, support_cols=F_.support_cols_of(p))
    assert int(cols2.max()) >= 32 and int(cols2.min()) >= 32
    assert F_.support_cols_of(p.clone()) is None and F_.support_cols_of(p.detach()) is None


@pytest.mark.parametrize("shape", [(1,), (7,), (640,), (64, 10), (100000,), (3000001,)])
def test_support_entropy_fused_reduction(dev, shape):
    """-(p @ log p)/B over the positive entries (structmarg.py:51-54, 200) as two launches forward
    and one backward: value and gradient against the reference's own torch expression in fp64."""
    from lvmhelpers import functional as F_
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    p = torch.rand(shape, device=dev, generator=g)
    p = torch.where(torch.rand(shape, device=dev, generator=g) < 0.3, torch.zeros_like(p), p)   # exact zeros
    p = (p / p.sum()).requires_grad_(True)
    inv_b = 1.0 / 64
    h = F_.support_entropy(p, inv_b)
    assert h.dim() == 0
    (3.0 * h).backward()
    p64 = p.detach().double().requires_grad_(True)
    pos = p64[p64 > 0]
    ref = -(pos @ torch.log(pos)) * inv_b
    (3.0 * ref).backward()
    assert abs(h.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
    gref = p64.grad
    assert torch.all(p.grad[p.detach() <= 0] == 0)
    assert torch.allclose(p.grad.double(), torch.where(p64 > 0, gref, torch.zeros_like(gref)), rtol=1e-5, atol=1e-7)


def test_support_entropy_of_nothing_is_zero(dev):
    from lvmhelpers import functional as F_
    assert F_.support_entropy(torch.zeros(5, device=dev)).item() == 0.0
    assert F_.support_entropy(torch.zeros(0, device=dev)).item() == 0.0


@pytest.mark.parametrize("R,K", [(64, 33), (1, 1), (300, 10), (1000, 256), (7, 4096)])
def test_gather_support_is_the_indexing_it_replaces(dev, R, K):
    """vals = p[rows, cols] without a launch (the compaction wrote them) and with ONE scatter as
    its backward: same values, same gradient as torch's advanced indexing (structmarg.py:123-126,
    189-190), rows without entries included."""
    from lvmhelpers import _native, functional as F_
    g = torch.Generator(device=dev).manual_seed(R * K)
    p = torch.rand(R, K, device=dev, generator=g)
    p = torch.where(torch.rand(R, K, device=dev, generator=g) < 0.8, torch.zeros_like(p), p)
    if R > 2:
        p[1] = 0.0                                     # an empty row
    off, rows, cols, vals = _native.support_compact(p)
    pa = p.clone().requires_grad_(True)
    pb = p.clone().requires_grad_(True)
    va = F_.gather_support(pa, off, cols, vals)
    vb = pb[rows.long(), cols.long()]
    assert torch.equal(va, vb)
    w = torch.randn(va.shape[0], device=dev, generator=g)
    (va * w).sum().backward()
    (vb * w).sum().backward()
    assert torch.equal(pa.grad, pb.grad)
    dense = _native.support_scatter(w, off, cols, R, K)
    assert torch.equal(dense, pb.grad)


@pytest.mark.parametrize("R,inner,dens", [(64, (16, 256), 0.03), (7, (5,), 0.5), (300, (33,), 0.1), (1, (8,), 1.0),
                                          (40, (4, 3, 2), 0.0)])
def test_segment_sum_is_the_gather_backward(dev, oracle_mod, R, inner, dens):
    """smlvm_support_segment_sum / functional.gather_rows: per-row sums of one vector per compacted
    entry == the backward of table[rows] (index_add), with empty rows, and with static-shape padding
    behind the N real entries ignored; and it is the same to the last bit on every call."""
    from lvmhelpers import _native, functional as F_
    rng = np.random.default_rng(R + len(inner))
    p = rng.random((R, 12)).astype(np.float32)
    p[rng.random((R, 12)) >= dens] = 0.0
    off, rows, _, _ = oracle_mod.support_compact(p)
    N, pad = rows.shape[0], 5
    g = rng.standard_normal((N + pad,) + inner).astype(np.float32)
    want = np.zeros((R,) + inner, np.float64)
    np.add.at(want, rows, g[:N].astype(np.float64))
    offt = torch.from_numpy(off).to(dev)
    got = _native.support_segment_sum(torch.from_numpy(g).to(dev), offt, R)
    assert tuple(got.shape) == (R,) + inner
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    assert torch.equal(got, _native.support_segment_sum(torch.from_numpy(g).to(dev), offt, R))
    # as an autograd function
    table = torch.randn((R,) + inner, device=dev, requires_grad=True)
    rows_t = torch.from_numpy(np.concatenate([rows, np.zeros(pad, np.int32)])).to(dev).long()
    w = torch.from_numpy(g).to(dev)
    w[N:] = 0.0                                               # (padding carries no gradient)
    (F_.gather_rows(table, rows_t, offt) * w).sum().backward()
    ref = torch.randn(1)  # placeholder to keep names distinct
    t2 = table.detach().clone().requires_grad_(True)
    (t2.index_select(0, rows_t) * w).sum().backward()
    assert float((table.grad - t2.grad).abs().max()) <= 1e-5 * max(1.0, float(t2.grad.abs().max()))


@pytest.mark.parametrize("R,K,sigma", [(700, 256, 0.05), (300, 40, 0.02), (513, 128, 0.05), (200, 512, 0.03),
                                       (90, 1000, 0.02), (64, 1024, 0.05), (1000, 256, 1.0), (257, 100, 0.3)])
def test_fused_compaction_counts_unlisted_rows_from_the_patterns(dev, oracle_mod, R, K, sigma):
    """smlvm_support_compact_fused_masks: rows whose list overflowed are counted from the forward's
    16-bit-per-lane patterns instead of from p -- same offsets / rows / cols / vals as the oracle,
    for every tile shape that writes patterns, ragged R, listed and unlisted rows side by side."""
    from lvmhelpers import _native, functional as F_
    rng = np.random.default_rng(R + K)
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    x[::7] *= 40.0                                        # some rows with short supports (listed)
    x[3] = 0.5                                            # an all-equal row (K non-zeros)
    xt = torch.from_numpy(x).to(dev)
    out = _native.sparsemax_fwd(xt, want_cols=True, want_mask=True, want_vals=True)
    p, cols, mask, vals = out[0], out[5], out[6], out[7]
    assert mask is not None
    unlisted = int((cols[:, 0] < -1).sum())
    off_ref, r_ref, c_ref, v_ref = oracle_mod.support_compact(p.cpu().numpy())
    for vl in (None, vals):
        off, rws, cls, vs = _native.support_compact_fused(p, cols, vl, support_mask=mask)[:4]
        assert np.array_equal(off.cpu().numpy(), off_ref)
        assert np.array_equal(rws.cpu().numpy(), r_ref) and np.array_equal(cls.cpu().numpy(), c_ref)
        assert np.array_equal(vs.cpu().numpy(), v_ref)
    if sigma <= 0.05 and K >= 100:
        assert unlisted > R // 2                          # (the case this is for: narrow logits)
    # through the autograd function: the tag carries the patterns only when the forward wrote them
    pt = F_.sparsemax(xt, dim=-1)
    assert F_.support_mask_of(pt) is not None
    off, rws, cls, vs = _native.support_compact(pt.detach(), support_cols=F_.support_cols_of(pt),
                                                support_mask=F_.support_mask_of(pt))
    assert np.array_equal(off.cpu().numpy(), off_ref) and np.array_equal(cls.cpu().numpy(), c_ref)


def test_patterns_are_not_offered_when_the_forward_did_not_write_them(dev):
    from lvmhelpers import functional as F_
    g = torch.Generator(device=dev).manual_seed(0)
    base = torch.randn(50 * 256 + 1, device=dev, generator=g)
    x_unaligned = base[1:].view(50, 256)                  # 4-byte aligned rows: the scalar path
    assert x_unaligned.data_ptr() % 16 != 0
    assert F_.support_mask_of(F_.sparsemax(x_unaligned, dim=-1)) is None
    assert F_.support_mask_of(F_.sparsemax(torch.randn(50, 255, device=dev), dim=-1)) is None   # K % 4 != 0
    assert F_.support_mask_of(F_.sparsemax(torch.randn(50, 2048, device=dev), dim=-1)) is None  # bitmaps, not patterns
    assert F_.support_mask_of(F_.sparsemax(torch.randn(50, 10, device=dev), dim=-1)) is None    # no mask buffer
    assert F_.support_mask_of(F_.sparsemax(torch.randn(50, 256, device=dev), dim=-1)) is not None
