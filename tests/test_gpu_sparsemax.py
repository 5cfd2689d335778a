"""GPU parity of kernel 1 (sparsemax fwd/bwd, fused entropy/nnz) against the CPU oracle.

Bar: support (p > 0 pattern) and supp_size bit-exact; tau and p bit-exact (the kernel reproduces the
oracle's fp32 expressions from an fp64 candidate sum); gradients within 1e-6 relative of the row
scale (summation order of the masked row sum differs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 10), (64, 256), (7, 3), (5, 1), (33, 31), (64, 32), (19, 100), (16, 128), (130, 200),
          (9, 512), (6, 1000), (5, 1024), (4, 1500), (3, 2048), (3, 4096), (3, 4100), (2, 8192),
          (2, 10000), (2, 16384), (2, 32768), (2, 65536), (1, 50001)]


def _cand_count(xrow, K):
    """Elements the kernel's candidate lists have to hold: those above csrc/sparsemax.hip
    cand_floor(K) in the max-shifted row (fp32 arithmetic as in the kernel)."""
    f = np.float32
    a = f(3.0) * f(K) * f(2.0 ** -23)
    floor = -((f(1.0) + f(2.0 ** -23)) / (f(1.0) - a)) * (f(1.0) + f(1e-5))
    return np.count_nonzero((xrow - xrow.max()) > floor)


def _rows(rng, R, K, kind):
    if kind == "normal":
        return rng.standard_normal((R, K)).astype(np.float32)
    if kind == "narrow":
        return (0.05 * rng.standard_normal((R, K))).astype(np.float32)
    if kind == "wide":
        return (10 * rng.standard_normal((R, K))).astype(np.float32)
    if kind == "adversarial":
        x = rng.standard_normal((R, K)).astype(np.float32)
        for r in range(R):
            m = r % 6
            if m == 0:
                x[r] = 0.25                      # all equal -> support K
            elif m == 1:
                x[r] = -1.0; x[r, r % K] = 3.0   # one dominant logit -> support 1
            elif m == 2:
                x[r] = -1.0; x[r, 0] = 0.0       # exact tie at the threshold (tau = -1)
            elif m == 3:
                x[r] = np.round(x[r] * 4) / 4    # many exact ties
            elif m == 4:
                x[r, : max(1, K // 2)] = 1.0     # half the row tied at the top
            # m == 5: plain random row
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["normal", "narrow", "wide", "adversarial"])
@pytest.mark.parametrize("R,K", SHAPES)
def test_fwd_matches_oracle(dev, oracle_mod, R, K, kind):
    from lvmhelpers import _native
    rng = np.random.default_rng(1000 * K + R)
    x = _rows(rng, R, K, kind)
    p_ref, s_ref, tau_ref, nnz_ref = oracle_mod.sparsemax_fwd(x)
    ent_ref = oracle_mod.entropy(p_ref)
    xt = torch.from_numpy(x).to(dev)
    for extra in (False, True):
        p, supp, nnz, tau, ent = _native.sparsemax_fwd(xt, want_nnz=extra, want_tau=extra,
                                                       want_entropy=extra)
        p = p.cpu().numpy()
        assert np.array_equal(p > 0, p_ref > 0), "support pattern differs"
        assert np.array_equal(supp.cpu().numpy(), s_ref)
        assert np.array_equal(p, p_ref), f"max |dp| = {np.abs(p - p_ref).max()}"
        if extra:
            assert np.array_equal(nnz.cpu().numpy(), nnz_ref)
            assert np.array_equal(tau.cpu().numpy(), tau_ref)
            np.testing.assert_allclose(ent.cpu().numpy(), ent_ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("R,K", SHAPES)
def test_unaligned_rows_use_scalar_path(dev, oracle_mod, R, K):
    """A view whose base pointer is not 16-byte aligned must take the scalar-load kernels."""
    from lvmhelpers import _native
    rng = np.random.default_rng(K)
    x = rng.standard_normal((R, K)).astype(np.float32)
    buf = torch.empty(R * K + 1, dtype=torch.float32, device=dev)
    xt = buf[1:].view(R, K)
    xt.copy_(torch.from_numpy(x))
    assert xt.data_ptr() % 16 != 0
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(x)
    # _native.contiguous() keeps the view (it is contiguous), so the kernel sees the odd pointer
    p, supp, _, _, _ = _native.sparsemax_fwd(xt)
    assert np.array_equal(p.cpu().numpy(), p_ref)
    assert np.array_equal(supp.cpu().numpy(), s_ref)


@pytest.mark.parametrize("kind", ["normal", "narrow", "adversarial"])
@pytest.mark.parametrize("R,K", SHAPES)
def test_bwd_matches_oracle(dev, oracle_mod, R, K, kind):
    from lvmhelpers import _native
    rng = np.random.default_rng(7 * K + R)
    x = _rows(rng, R, K, kind)
    dy = rng.standard_normal((R, K)).astype(np.float32)
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(x)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    dx = _native.sparsemax_bwd(torch.from_numpy(dy).to(dev), torch.from_numpy(p_ref).to(dev),
                               torch.from_numpy(s_ref).to(dev)).cpu().numpy()
    assert np.array_equal(dx != 0, dx_ref != 0) or np.array_equal((dx == 0) | (p_ref != 0), np.ones_like(dx, bool))
    assert np.array_equal(dx[p_ref == 0], dx_ref[p_ref == 0])  # exact zeros off the support
    scale = np.abs(dy).max(axis=1, keepdims=True)
    assert np.all(np.abs(dx - dx_ref) <= 1e-6 * scale), np.abs(dx - dx_ref).max()


def test_autograd_function(dev, oracle_mod):
    from lvmhelpers import functional as F_
    rng = np.random.default_rng(3)
    x = rng.standard_normal((64, 256)).astype(np.float32)
    w = rng.standard_normal((64, 256)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    p = F_.sparsemax(xt, dim=-1)
    (p * torch.from_numpy(w).to(dev)).sum().backward()
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(x)
    dx_ref = oracle_mod.sparsemax_bwd(w, p_ref, s_ref)
    assert np.array_equal(p.detach().cpu().numpy(), p_ref)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), dx_ref, rtol=0, atol=1e-6 * np.abs(w).max())
    # dim != -1
    xt2 = torch.from_numpy(x).to(dev)
    p0 = F_.sparsemax(xt2, dim=0)
    p0_ref, _, _, _ = oracle_mod.sparsemax_fwd(np.ascontiguousarray(x.T))
    assert np.array_equal(p0.cpu().numpy(), p0_ref.T)


def test_large_sizes_properties(dev):
    """BASELINE sizes: size-independent properties (sum to one, shift invariance, threshold
    consistency, idempotent support) instead of the oracle."""
    from lvmhelpers import _native
    g = torch.Generator(device=dev).manual_seed(1234)
    for R, K in ((1 << 16, 256), (1 << 12, 4096), (256, 65536)):
        x = torch.randn(R, K, device=dev, generator=g)
        p, supp, nnz, tau, _ = _native.sparsemax_fwd(x, want_nnz=True, want_tau=True)
        assert torch.all(p >= 0)
        assert torch.all((p.double().sum(-1) - 1).abs() <= 1e-6 * max(1.0, K / 4096) + 2e-6)
        z = x - x.max(dim=-1, keepdim=True)[0]
        assert torch.equal(p > 0, z > tau.unsqueeze(-1))
        assert torch.equal(nnz, (p > 0).sum(-1).to(torch.int32))
        assert torch.equal(supp, nnz)
        p2, supp2, _, _, _ = _native.sparsemax_fwd(x + 3.0)
        assert torch.equal(supp2, supp)
        assert torch.all((p2 - p).abs() <= 4e-6)
        dy = torch.randn(R, K, device=dev, generator=g)
        dx = _native.sparsemax_bwd(dy, p, supp)
        assert torch.equal(dx != 0, (p != 0) & (dx != 0))
        assert torch.all(dx.double().sum(-1).abs() <= 1e-4)


@pytest.mark.parametrize("R,K,sigma", [(1 << 20, 4096, 1.0), (1 << 20, 256, 1.0), (1 << 15, 65536, 1.0),
                                       (1 << 18, 4096, 0.05)])
def test_full_size_launches(dev, oracle_mod, R, K, sigma):
    """BASELINE.json configs[2] at its real launch sizes (1M x 4096, 1M x 256, 32768 x 65536 per
    launch; plus a wide-support case that takes the streaming backward).  Whole-tensor checks are
    size-independent properties evaluated in row chunks; 1024 rows sampled across the launch are
    compared with the CPU oracle (p, supp bit-exact; dx within 1e-6 of the row scale)."""
    from lvmhelpers import _native
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.empty(R, K, device=dev).normal_(0.0, sigma, generator=g)
    dy = torch.empty(R, K, device=dev).normal_(0.0, 1.0, generator=g)
    p, supp, nnz, tau, _, cols = _native.sparsemax_fwd(x, want_nnz=True, want_tau=True, want_cols=True)
    dx_p = _native.sparsemax_bwd(dy, p, supp)           # reads p
    dx = _native.sparsemax_bwd_cols(dy, cols, p, supp)  # reads the support lists
    assert bool(((dx - dx_p).abs() <= 1e-6 * dy.abs().amax(dim=-1, keepdim=True)).all())
    del dx_p
    assert torch.equal(supp, nnz)
    step = max(1, (1 << 26) // K)                      # 256 MB chunks
    for lo in range(0, R, step):
        sl = slice(lo, min(R, lo + step))
        ps, dxs = p[sl], dx[sl]
        assert bool((ps >= 0).all())
        assert bool(((ps.sum(-1, dtype=torch.float64) - 1).abs() <= 3e-6).all())
        assert torch.equal((ps > 0).sum(-1).to(torch.int32), supp[sl])
        z = x[sl] - x[sl].max(dim=-1, keepdim=True)[0]
        assert torch.equal(ps > 0, z > tau[sl].unsqueeze(-1))
        assert bool(((dxs != 0) <= (ps != 0)).all())             # zero off the support
        assert bool((dxs.sum(-1, dtype=torch.float64).abs() <= 1e-4).all())
        del z
    idx = torch.randint(0, R, (1024,), device=dev, generator=g)
    idx[0], idx[-1] = 0, R - 1
    xs, dys = x[idx].cpu().numpy(), dy[idx].cpu().numpy()
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(xs)
    dx_ref = oracle_mod.sparsemax_bwd(dys, p_ref, s_ref)
    assert np.array_equal(p[idx].cpu().numpy(), p_ref)
    assert np.array_equal(supp[idx].cpu().numpy(), s_ref)
    scale = np.abs(dys).max(axis=1, keepdims=True)
    assert np.all(np.abs(dx[idx].cpu().numpy() - dx_ref) <= 1e-6 * scale)


@pytest.mark.parametrize("kind", ["normal", "narrow", "adversarial"])
@pytest.mark.parametrize("R,K", SHAPES)
def test_support_lists_and_list_backward(dev, oracle_mod, R, K, kind):
    """Forward by-product: the columns of p > 0 per row (any order, -1 padding, -2 = no list), and
    the backward that uses them instead of reading p.  The lists must name exactly the oracle's
    non-zeros; dx must match the oracle like the p-based backward does."""
    from lvmhelpers import _native
    rng = np.random.default_rng(7 * R + K)
    x = _rows(rng, R, K, kind)
    dy = rng.standard_normal((R, K)).astype(np.float32)
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(x)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    p, supp, _, _, _, cols = _native.sparsemax_fwd(torch.from_numpy(x).to(dev), want_cols=True)
    assert np.array_equal(p.cpu().numpy(), p_ref)
    c = cols.cpu().numpy()
    cap = c.shape[1]
    listed = 0
    for r in range(R):
        nz = np.flatnonzero(p_ref[r] > 0)
        if nz.size > cap:
            assert c[r, 0] in (-2, -3)
        elif K > 4096 and c[r, 0] == -2:
            # rows held by several waves list their support only when the threshold search ran on
            # the candidate list (<= 64 elements within 1 of the maximum per wave, <= 256 per row)
            assert _cand_count(x[r], K) > 64
        else:
            assert sorted(c[r, :nz.size].tolist()) == nz.tolist()
            assert np.all(c[r, nz.size:] == -1)
            listed += 1
    if kind == "normal":
        assert listed == R
    dx = _native.sparsemax_bwd_cols(torch.from_numpy(dy).to(dev), cols, p, supp).cpu().numpy()
    scale = np.maximum(np.abs(dy).max(axis=1, keepdims=True), 1e-30)
    assert np.all(np.abs(dx - dx_ref) <= 1e-6 * scale)
    assert np.array_equal(dx != 0, (p_ref != 0) & (dx != 0))


@pytest.mark.parametrize("sigma", [0.3, 0.1, 0.05, 0.02, 0.005])
@pytest.mark.parametrize("K", [1100, 2048, 3000, 4096, 8192, 20000, 65536])
def test_narrow_rows_lazy_candidates(dev, oracle_mod, K, sigma):
    """Rows with more than 256 elements within 1 of the maximum: the forward raises the bound with
    full-width Michelot steps and compacts its candidates above (bound - 1e-3).  Half of the rows
    get clusters planted around the oracle's threshold -- hundreds of elements inside the margin, a
    pair 1 ulp below tau, a pair of exact ties at tau -- and p, supp, tau and the support lists
    must still equal the oracle's bit for bit; the widest cases (support > 256) fall through to the
    full-width search.  (Hundreds to thousands of exact ties AT tau: test_tie_groups_at_threshold.)"""
    from lvmhelpers import _native
    R = 256 if K <= 4096 else 48
    rng = np.random.default_rng(int(1e4 * sigma) + K)
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    _, s0, tau0, _ = oracle_mod.sparsemax_fwd(x)
    mx = x.max(axis=1)
    for r in range(0, R, 2):
        t = np.float32(tau0[r] + mx[r])                 # threshold in the row's own frame
        below = np.flatnonzero(x[r] < t - 2e-3)          # far from the support: free to move
        rng.shuffle(below)
        n = min(below.size, 400)
        k = r % 8
        if k == 0:
            x[r, below[:n]] = t - np.float32(5e-4)       # a crowd inside the margin
        elif k == 2:
            x[r, below[:2]] = np.nextafter(t, np.float32(-np.inf))   # 1 ulp below tau
            x[r, below[2:n]] = t - rng.uniform(1e-5, 2e-3, n - 2).astype(np.float32)
        elif k == 4:
            x[r, below[:2]] = t                          # exact ties at tau (outside the support)
            x[r, below[2:n]] = t - np.float32(1.5e-3)
        else:
            x[r, below[:n]] = t - rng.uniform(1e-5, 2e-3, n).astype(np.float32)
    p_ref, s_ref, tau_ref, _ = oracle_mod.sparsemax_fwd(x)
    dy = rng.standard_normal((R, K)).astype(np.float32)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    xt = torch.from_numpy(x).to(dev)
    p, supp, _, tau, _, cols = _native.sparsemax_fwd(xt, want_tau=True, want_cols=True)
    assert np.array_equal(supp.cpu().numpy(), s_ref)
    assert np.array_equal(tau.cpu().numpy(), tau_ref)
    assert np.array_equal(p.cpu().numpy(), p_ref)
    c = cols.cpu().numpy()
    cap = c.shape[1]
    for r in range(R):
        nz = np.flatnonzero(p_ref[r] > 0)
        if nz.size > cap:
            assert c[r, 0] in (-2, -3)
        elif K <= 4096 or c[r, 0] != -2:   # wide rows may keep no list (per-wave candidate limit)
            assert sorted(c[r, :nz.size].tolist()) == nz.tolist() and np.all(c[r, nz.size:] == -1)
    dx = _native.sparsemax_bwd_cols(torch.from_numpy(dy).to(dev), cols, p, supp).cpu().numpy()
    assert np.all(np.abs(dx - dx_ref) <= 1e-6 * np.abs(dy).max(axis=1, keepdims=True))
    # without the lists (different kernel instantiation of the same search)
    p2, supp2, _, _, _ = _native.sparsemax_fwd(xt)
    assert np.array_equal(p2.cpu().numpy(), p_ref) and np.array_equal(supp2.cpu().numpy(), s_ref)


def _tie_rows(rng, K, oracle_mod):
    """Rows whose oracle threshold sits inside / next to big groups of tied values: there the
    oracle's fp32 rank test is decided by rounding, passes at isolated ranks, and its support size
    (the COUNT of passing ranks) is not the size of a prefix.  One row per construction."""
    f32 = np.float32
    rows = []

    def base(sig):
        x = (sig * rng.standard_normal(K)).astype(f32)
        _, _, tau0, _ = oracle_mod.sparsemax_fwd(x[None])
        t = f32(tau0[0] + x.max())
        below = np.flatnonzero(x < t)
        rng.shuffle(below)
        return x, t, below

    for sig in (1.0, 0.05):
        for n in (300, K // 4, K - 64):
            for mode in range(5):
                x, t, below = base(sig)
                idx = below[:min(n, below.size)]
                if mode == 0:
                    x[idx] = t                                       # ties exactly at tau
                elif mode == 1:
                    x[idx] = np.nextafter(t, f32(-np.inf))           # 1 ulp below
                elif mode == 2:
                    x[idx] = np.nextafter(t, f32(np.inf))            # 1 ulp above
                elif mode == 3:                                      # three groups an ulp apart
                    h = idx.size // 3
                    x[idx[:h]] = t
                    x[idx[h:2 * h]] = np.nextafter(t, f32(-np.inf))
                    x[idx[2 * h:]] = np.nextafter(t, f32(np.inf))
                else:
                    x[idx] = t - f32(1e-5)                           # a crowd 1e-5 below tau
                rows.append(x)
    for eps in (0.0, 1e-6, -1e-6, 2e-3, -2e-3, 1e-4, 2e-2, 1e-5, -1e-5):
        x = np.zeros(K, f32)                                         # one spike, K-1 ties at spike-1-eps
        x[7 % K] = f32(1.0 + eps)
        rows.append(x)
    for a in (2, 100, K // 2):
        for d in (1.0 / a, 1.0 / a + 1e-6, 1.0 / a - 1e-6):
            x = np.full(K, f32(-d), f32)                             # a at 0, K-a at the a-element tau
            x[:a] = 0.0
            rows.append(x)
    return np.stack(rows)


@pytest.mark.parametrize("K", [256, 1000, 4096, 65536, 131072])
def test_tie_groups_at_threshold(dev, oracle_mod, K):
    """Hundreds to 131k elements tied exactly at, one ulp around, or 1e-5 below the threshold: supp
    (the oracle's count of passing ranks, even where they are not a prefix), tau, p and the lists
    must equal the oracle's bit for bit, with and without the support lists, and the backward that
    divides by that supp must agree."""
    from lvmhelpers import _native
    rng = np.random.default_rng(K)
    x = _tie_rows(rng, K, oracle_mod)
    R = x.shape[0]
    p_ref, s_ref, tau_ref, nnz_ref = oracle_mod.sparsemax_fwd(x)
    assert np.any(s_ref != nnz_ref)          # the construction does split tie groups
    xt = torch.from_numpy(x).to(dev)
    p, supp, nnz, tau, _, cols, mask = _native.sparsemax_fwd(xt, want_nnz=True, want_tau=True,
                                                             want_cols=True, want_mask=True)
    bad = np.flatnonzero(supp.cpu().numpy() != s_ref)
    assert bad.size == 0, (bad[:8], supp.cpu().numpy()[bad[:8]], s_ref[bad[:8]])
    assert np.array_equal(tau.cpu().numpy(), tau_ref)
    assert np.array_equal(p.cpu().numpy(), p_ref)
    assert np.array_equal(nnz.cpu().numpy(), nnz_ref)
    p2, supp2, _, _, _ = _native.sparsemax_fwd(xt)
    assert np.array_equal(p2.cpu().numpy(), p_ref) and np.array_equal(supp2.cpu().numpy(), s_ref)
    c = cols.cpu().numpy()
    cap = c.shape[1]
    for r in range(R):
        nz = np.flatnonzero(p_ref[r] > 0)
        if c[r, 0] >= -1:
            assert sorted(c[r, :nz.size].tolist()) == nz.tolist() and np.all(c[r, nz.size:] == -1)
        elif K <= 4096:
            assert nz.size > cap
    dy = rng.standard_normal((R, K)).astype(np.float32)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    dyt = torch.from_numpy(dy).to(dev)
    scale = np.abs(dy).max(axis=1, keepdims=True)
    for dxt in (_native.sparsemax_bwd_cols(dyt, cols, p, supp, mask), _native.sparsemax_bwd(dyt, p, supp)):
        assert np.all(np.abs(dxt.cpu().numpy() - dx_ref) <= 2e-6 * scale)


@pytest.mark.parametrize("K", [256, 1000, 4096, 5000, 8192, 16384, 50001, 65536])
def test_few_valued_rows_and_exact_tie_groups(dev, oracle_mod, K):
    """Round 3 shortcuts of the threshold search, bit for bit against the oracle:
      * tie groups whose rank tests are exact in fp32 (integer / dyadic values: the group passes or
        fails as a whole, nothing is counted rank by rank) next to groups that are NOT exact
        (values off a dyadic grid by an ulp, long groups whose sums leave the 24-bit range);
      * rows made of one or two distinct values (closed form in the workgroup kernels, K > 4096)
        and the rows that only look like them: every lane holds a zero but other values exist, a
        third value below the plateau, -inf entries, plateaus at and around 1 below the maximum."""
    from lvmhelpers import _native
    rng = np.random.default_rng(K + 5)
    f32 = np.float32
    rows = []

    def two_valued(a, v, base=0.0):
        x = np.full(K, f32(base + v), f32)
        x[rng.choice(K, a, replace=False)] = f32(base)
        return x

    for v in (-1.0, -0.5, -0.25, -0.75, -1.0 + 2 ** -20, -1.0 - 2 ** -20, -0.999, -1e-3, -2.0, -1.0 / 3, -2 ** -10):
        for a in (1, 2, 3, 7, min(64, K - 1), K // 2, K - 1):
            rows.append(two_valued(a, v))
    rows.append(two_valued(1, -1.0, base=3.0))
    rows.append(two_valued(1, -1.0, base=-7.25))
    rows.append(np.full(K, f32(0.25), f32))                       # all equal
    rows.append(np.full(K, f32(-3.0), f32))
    x = np.zeros(K, f32); x[1::2] = -0.5; x[3] = -0.75; rows.append(x)                 # a third value
    x = two_valued(1, -1.0); x[K // 3] = f32(-1.5); rows.append(x)                     # ... below the plateau
    x = two_valued(5, -1.0); x[K // 2] = -np.inf; rows.append(x)                       # -inf entry
    x = np.full(K, f32(0.0), f32); x[K - 1] = -np.inf; rows.append(x)
    x = np.zeros(K, f32); x[rng.choice(K, K // 2, replace=False)] = f32(-1e-4); rows.append(x)
    # exact and inexact tie groups under a few distinct larger values (the wave kernels' edge walk)
    for v, top in ((-0.5, (0.0, -0.125)), (-0.25, (0.0,)), (-0.375, (0.0, -0.25, -0.3125)), (-0.3, (0.0, -0.1))):
        x = np.full(K, f32(v), f32)
        x[:len(top)] = np.array(top, f32)
        rows.append(rng.permutation(x))
    x = np.stack(rows)
    p_ref, s_ref, tau_ref, nnz_ref = oracle_mod.sparsemax_fwd(x)
    xt = torch.from_numpy(x).to(dev)
    p, supp, nnz, tau, _, cols, mask = _native.sparsemax_fwd(xt, want_nnz=True, want_tau=True, want_cols=True,
                                                             want_mask=True)
    bad = np.flatnonzero(supp.cpu().numpy() != s_ref)
    assert bad.size == 0, (bad[:8], supp.cpu().numpy()[bad[:8]], s_ref[bad[:8]])
    assert np.array_equal(tau.cpu().numpy(), tau_ref)
    assert np.array_equal(p.cpu().numpy(), p_ref)
    assert np.array_equal(nnz.cpu().numpy(), nnz_ref)
    p2, supp2, _, _, _ = _native.sparsemax_fwd(xt)
    assert np.array_equal(p2.cpu().numpy(), p_ref) and np.array_equal(supp2.cpu().numpy(), s_ref)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    dxt = _native.sparsemax_bwd_cols(torch.from_numpy(dy).to(dev), cols, p, supp, mask)
    assert np.all(np.abs(dxt.cpu().numpy() - dx_ref) <= 2e-6 * np.abs(dy).max(axis=1, keepdims=True))


@pytest.mark.parametrize("K", [64, 72, 100, 128, 129, 200, 256, 257, 300, 512])
@pytest.mark.parametrize("cap", [8, 32, 64])
def test_few_valued_rows_inside_mixed_waves_of_short_rows(dev, oracle_mod, K, cap):
    """Rows of K <= 512 share a wave (2-8 rows side by side).  One- and two-valued rows are solved in
    closed form there and kept out of the wave's candidate lists, so that their neighbours still get
    the short search; every kind next to every other kind, in random order: p / supp / tau / nnz bit
    for bit, the support lists exactly the non-zeros (or the overflow mark), the list backward and
    the value lists."""
    from lvmhelpers import _native
    rng = np.random.default_rng(31 * K + cap)
    f32 = np.float32
    rows = []

    def two_valued(a, v, base=0.0):
        x = np.full(K, f32(base + v), f32)
        x[rng.choice(K, a, replace=False)] = f32(base)
        return x

    for rep in range(6):
        for v in (-1.0, -0.5, -0.999, -1.0 - 2 ** -20, -1.0 + 2 ** -20, -1e-3, -2.0, -1.0 / 3, -1.0 / 7):
            for a in (1, 2, 5, cap, cap + 1, min(33, K - 1), K // 2, K - 1):
                rows.append(two_valued(min(a, K - 1), v, base=float(rng.choice([0.0, 0.25, -3.0]))))
        rows.append(np.full(K, f32(rng.choice([0.25, 0.0, -3.0])), f32))
        x = two_valued(3, -1.0); x[rng.integers(K)] = -np.inf; rows.append(x)
        x = np.full(K, f32(0.5), f32); x[rng.integers(K)] = -np.inf; rows.append(x)
        x = two_valued(1, -1.0); x[K // 3] = f32(-1.5); rows.append(x)          # a third value
        x = np.zeros(K, f32); x[1::2] = -0.5; x[3] = -0.75; rows.append(x)
    n_few = len(rows)
    rows += list(rng.standard_normal((n_few, K)).astype(f32))
    rows += list((0.05 * rng.standard_normal((n_few // 2, K))).astype(f32))
    rows += list((3.0 * rng.standard_normal((n_few // 2, K))).astype(f32))
    x = np.stack(rows)[rng.permutation(len(rows))]
    R = x.shape[0]
    p_ref, s_ref, tau_ref, nnz_ref = oracle_mod.sparsemax_fwd(x)
    xt = torch.from_numpy(x).to(dev)
    for want_cols in (True, False):
        out = _native.sparsemax_fwd(xt, want_nnz=True, want_tau=True, want_cols=want_cols, want_mask=want_cols,
                                    **({"cols_cap": cap} if want_cols else {}))
        p, supp, nnz, tau = out[0], out[1], out[2], out[3]
        bad = np.flatnonzero(supp.cpu().numpy() != s_ref)
        assert bad.size == 0, (bad[:8], supp.cpu().numpy()[bad[:8]], s_ref[bad[:8]])
        assert np.array_equal(tau.cpu().numpy(), tau_ref)
        assert np.array_equal(p.cpu().numpy(), p_ref)
        assert np.array_equal(nnz.cpu().numpy(), nnz_ref)
        if not want_cols:
            continue
        cols, mask = out[5], out[6]
        c = cols.cpu().numpy()
        assert c.shape[1] == cap
        for r in range(R):
            nz = np.flatnonzero(p_ref[r] > 0)
            if nz.size > cap:
                assert c[r, 0] in (-2, -3), (r, nz.size, c[r, :4])
            else:
                assert sorted(c[r, :nz.size].tolist()) == nz.tolist(), (r, nz.size, c[r, :8])
                assert np.all(c[r, nz.size:] == -1)
        dy = rng.standard_normal(x.shape).astype(np.float32)
        dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
        dxt = _native.sparsemax_bwd_cols(torch.from_numpy(dy).to(dev), cols, p, supp, mask)
        assert np.all(np.abs(dxt.cpu().numpy() - dx_ref) <= 2e-6 * np.abs(dy).max(axis=1, keepdims=True))
    # value lists next to the column lists (smlvm_sparsemax_fwd_lists)
    out = _native.sparsemax_fwd(xt, want_cols=True, want_vals=True, cols_cap=cap)
    c, v = out[5].cpu().numpy(), out[6].cpu().numpy()
    for r in range(R):
        n = int((p_ref[r] > 0).sum())
        if n <= cap:
            assert np.array_equal(v[r, :n], p_ref[r, c[r, :n]])


def test_mask_mark_follows_the_nonzero_count(dev, oracle_mod):
    """A row is marked -3 (support mask) exactly when #{p > 0} exceeds the list capacity; the
    oracle's supp_size can be smaller than that count when a tie group straddles the threshold
    (supp <= cap < nnz), and the mask must have been written for such a row: the masked backward
    has to equal the p-based one."""
    from lvmhelpers import _native
    f32 = np.float32
    K, cap = 4096, _native.SUPPORT_COLS_CAP
    rng = np.random.default_rng(5)
    rows = []
    # a support of m distinct values + a tie group of t values placed so that the group's common
    # value is a hair above the threshold of (m + some of the group): search for supp <= cap < nnz
    for trial in range(4000):
        m = int(rng.integers(1, cap))
        t = int(rng.integers(cap - m + 1, 3 * cap))
        top = np.sort(rng.uniform(-0.5, 0.0, m)).astype(f32)[::-1].copy()
        top[0] = 0.0
        v = f32((top.astype(np.float64).sum() - 1.0) / m) + f32(rng.choice([0.0, 1e-7, -1e-7, 3e-8]))
        x = np.full(K, f32(-5.0), f32)
        x[:m] = top
        x[m:m + t] = v
        _, s1, _, n1 = oracle_mod.sparsemax_fwd(x[None])
        if s1[0] <= cap < n1[0]:
            rows.append(rng.permutation(x))
            if len(rows) >= 8:
                break
    assert rows, "no supp <= cap < nnz row found"
    x = np.stack(rows)
    p_ref, s_ref, _, nnz_ref = oracle_mod.sparsemax_fwd(x)
    xt = torch.from_numpy(x).to(dev)
    dyt = torch.randn(x.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    p, supp, _, _, _, cols, mask = _native.sparsemax_fwd(xt, want_cols=True, want_mask=True)
    assert np.array_equal(p.cpu().numpy(), p_ref) and np.array_equal(supp.cpu().numpy(), s_ref)
    assert np.all(cols[:, 0].cpu().numpy() == -3)
    dx = _native.sparsemax_bwd_cols(dyt, cols, p, supp, mask)
    assert torch.equal(dx, _native.sparsemax_bwd(dyt, p, supp))
    # ... and it really comes from the mask: a masked row (<= 96 non-zeros) never reads p -- the
    # same call with p replaced by NaNs gives the same dx --, while a mask that was NOT written for
    # these rows (all bits set) gives something else
    nnz = (p > 0).sum(1)
    assert int(nnz.max()) <= 96
    dx_nan_p = _native.sparsemax_bwd_cols(dyt, cols, torch.full_like(p, float("nan")), supp, mask)
    assert torch.equal(dx_nan_p, dx)
    dx_junk = _native.sparsemax_bwd_cols(dyt, cols, p, supp, torch.full_like(mask, -1))
    assert not torch.equal(dx_junk, dx)


@pytest.mark.parametrize("R,K", [(6, 5000), (5, 8192), (4, 50001), (7, 4097), (9, 3001), (64, 255)])
def test_list_pair_on_rows_that_are_not_16_byte_addressable(dev, oracle_mod, R, K):
    """Odd K and/or a base pointer off by one float: the scalar-layout kernels (raw buffer accesses
    with hardware range checking for K > 4096) must give the same lists, p and dx."""
    from lvmhelpers import _native
    rng = np.random.default_rng(31 * R + K)
    x = rng.standard_normal((R, K)).astype(np.float32)
    dy = rng.standard_normal((R, K)).astype(np.float32)
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(x)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)

    def off_by_one(a):   # same values in a buffer whose rows start 4 bytes off a 16-byte boundary
        buf = torch.empty(a.size + 1, dtype=torch.float32, device=dev)
        v = buf[1:].view(a.shape)
        v.copy_(torch.from_numpy(a))
        assert v.data_ptr() % 16 != 0 and v.is_contiguous()
        return v

    xt, dyt = off_by_one(x), off_by_one(dy)
    p, supp, _, _, _, cols = _native.sparsemax_fwd(xt, want_cols=True)
    assert np.array_equal(p.cpu().numpy(), p_ref) and np.array_equal(supp.cpu().numpy(), s_ref)
    c = cols.cpu().numpy()
    for r in range(R):
        nz = np.flatnonzero(p_ref[r] > 0)
        assert nz.size <= c.shape[1]
        assert sorted(c[r, :nz.size].tolist()) == nz.tolist() and np.all(c[r, nz.size:] == -1)
    scale = np.abs(dy).max(axis=1, keepdims=True)
    for dxt in (_native.sparsemax_bwd_cols(dyt, cols, p, supp), _native.sparsemax_bwd(dyt, p, supp)):
        assert np.all(np.abs(dxt.cpu().numpy() - dx_ref) <= 1e-6 * scale)


@pytest.mark.parametrize("K,sigma", [(4096, 0.05), (4096, 0.02), (3000, 0.05), (2048, 0.03), (1028, 0.05),
                                     (4096, 0.005), (4096, 1.0)])
def test_support_mask_backward(dev, oracle_mod, K, sigma):
    """Rows with more non-zeros than a list holds: with the support-mask buffer the forward marks
    them -3 and leaves their non-zero pattern, and the backward gathers dy at those bits instead of
    reading p.  Same numbers in the same order as the p-based gather: dx must be bit-identical to
    it, and within 1e-6 of the oracle; without the buffer the mark stays -2."""
    from lvmhelpers import _native
    R = 300
    rng = np.random.default_rng(K + int(1e4 * sigma))
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    x[::7] *= 20                      # some rows with a short support: they keep their lists
    dy = rng.standard_normal((R, K)).astype(np.float32)
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(x)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    xt, dyt = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    p, supp, _, _, _, cols, mask = _native.sparsemax_fwd(xt, want_cols=True, want_mask=True)
    assert mask is not None and np.array_equal(p.cpu().numpy(), p_ref)
    head = cols[:, 0].cpu().numpy()
    nnz = (p_ref > 0).sum(1)
    cap = cols.shape[1]
    assert np.all(head[nnz > cap] == -3) and np.all(head[nnz <= cap] >= 0)
    dx_mask = _native.sparsemax_bwd_cols(dyt, cols, p, supp, mask)
    dx_p = _native.sparsemax_bwd(dyt, p, supp)
    assert torch.equal(dx_mask, dx_p)
    scale = np.abs(dy).max(axis=1, keepdims=True)
    assert np.all(np.abs(dx_mask.cpu().numpy() - dx_ref) <= 1e-6 * scale)
    # p is not read for the masked rows (gathered up to 96 non-zeros, dy streamed beyond): poison it
    poisoned = p.clone()
    poisoned[torch.from_numpy(nnz > cap).to(dev)] = float("nan")
    assert torch.equal(_native.sparsemax_bwd_cols(dyt, cols, poisoned, supp, mask), dx_mask)
    # without the buffer: plain overflow marks, p-based backward
    _, _, _, _, _, cols2 = _native.sparsemax_fwd(xt, want_cols=True)
    h2 = cols2[:, 0].cpu().numpy()
    assert np.all(h2[nnz > cap] == -2)
    assert torch.equal(_native.sparsemax_bwd_cols(dyt, cols2, p, supp), dx_p)


@pytest.mark.parametrize("R,K,sigma", [(301, 256, 0.05), (64, 256, 0.02), (77, 128, 0.05), (50, 64, 0.02),
                                       (99, 512, 0.05), (130, 1000, 0.05), (65, 1024, 0.03), (40, 36, 0.01),
                                       (123, 256, 1.0), (33, 260, 0.05)])
def test_support_patterns_of_short_rows(dev, oracle_mod, R, K, sigma):
    """32 < K <= 1024 (tiles of 16 registers, several rows per wave up to K = 512): with the mask
    buffer every row leaves its non-zero pattern; rows that overflow their list are marked -3, the
    others keep their lists; a wave of the backward that holds a -3 row takes all of its rows from
    the patterns and streams dy without reading p.  dx bit-identical to the p-based backward."""
    from lvmhelpers import _native
    rng = np.random.default_rng(R * 13 + K)
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    x[::5] *= 40                      # short supports in between: mixed waves
    dy = rng.standard_normal((R, K)).astype(np.float32)
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(x)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    xt, dyt = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    p, supp, _, _, _, cols, mask = _native.sparsemax_fwd(xt, want_cols=True, want_mask=True)
    assert mask is not None and np.array_equal(p.cpu().numpy(), p_ref)
    head = cols[:, 0].cpu().numpy()
    nnz = (p_ref > 0).sum(1)
    cap = cols.shape[1]
    assert np.all(head[nnz > cap] == -3) and np.all(head[nnz <= cap] >= 0)
    if sigma < 1.0:
        assert (nnz > cap).any() and (nnz <= cap).any()
    dx_mask = _native.sparsemax_bwd_cols(dyt, cols, p, supp, mask)
    dx_p = _native.sparsemax_bwd(dyt, p, supp)
    assert torch.equal(dx_mask, dx_p)
    scale = np.abs(dy).max(axis=1, keepdims=True)
    assert np.all(np.abs(dx_mask.cpu().numpy() - dx_ref) <= 1e-6 * scale)
    # p is not read for rows marked -3: poison it there
    poisoned = p.clone()
    poisoned[torch.from_numpy(nnz > cap).to(dev)] = float("nan")
    assert torch.equal(_native.sparsemax_bwd_cols(dyt, cols, poisoned, supp, mask), dx_mask)
    # the lists of the rows that kept them are what the plain forward writes
    _, _, _, _, _, cols2 = _native.sparsemax_fwd(xt, want_cols=True)
    keep = torch.from_numpy(nnz <= cap).to(dev)
    assert torch.equal(cols[keep], cols2[keep])
    assert np.all(cols2[:, 0].cpu().numpy()[nnz > cap] == -2)
    assert torch.equal(_native.sparsemax_bwd_cols(dyt, cols2, p, supp), dx_p)
    # compaction fed with the tagged lists: -3 rows are scanned from p
    off0, rows0, cols0, vals0 = _native.support_compact(p)
    off1, rows1, cols1, vals1 = _native.support_compact(p, support_cols=cols)
    assert torch.equal(off0, off1) and torch.equal(cols0, cols1) and torch.equal(vals0, vals1)


@pytest.mark.parametrize("kind", ["normal", "narrow", "verynarrow", "adversarial"])
@pytest.mark.parametrize("R,K", [(3, 65540), (2, 100000), (2, 131072), (2, 200001), (1, 262144)])
def test_rows_wider_than_a_workgroup(dev, oracle_mod, R, K, kind):
    """K > 65536: the row does not fit one workgroup's registers and is streamed (three reads out of
    HBM / cache, one write).  Same bar as everywhere: p, supp, tau bit-exact, lists exact, dx within
    1e-6 of the row scale, for both backward entry points."""
    from lvmhelpers import _native
    rng = np.random.default_rng(K + R)
    if kind == "verynarrow":          # support of several hundred: the whole search runs streaming
        x = (0.002 * rng.standard_normal((R, K))).astype(np.float32)
    else:
        x = _rows(rng, R, K, kind)
    dy = rng.standard_normal((R, K)).astype(np.float32)
    p_ref, s_ref, tau_ref, nnz_ref = oracle_mod.sparsemax_fwd(x)
    ent_ref = oracle_mod.entropy(p_ref)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    xt, dyt = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    p, supp, nnz, tau, ent, cols = _native.sparsemax_fwd(xt, want_nnz=True, want_tau=True,
                                                         want_entropy=True, want_cols=True)
    assert np.array_equal(supp.cpu().numpy(), s_ref)
    assert np.array_equal(tau.cpu().numpy(), tau_ref)
    assert np.array_equal(p.cpu().numpy(), p_ref)
    assert np.array_equal(nnz.cpu().numpy(), nnz_ref)
    np.testing.assert_allclose(ent.cpu().numpy(), ent_ref, rtol=1e-6, atol=1e-7)
    p2, supp2, _, _, _ = _native.sparsemax_fwd(xt)
    assert np.array_equal(p2.cpu().numpy(), p_ref) and np.array_equal(supp2.cpu().numpy(), s_ref)
    c = cols.cpu().numpy()
    for r in range(R):
        nz = np.flatnonzero(p_ref[r] > 0)
        if c[r, 0] != -2:
            assert nz.size <= c.shape[1]
            assert sorted(c[r, :nz.size].tolist()) == nz.tolist() and np.all(c[r, nz.size:] == -1)
        else:
            assert nz.size > c.shape[1] or _cand_count(x[r], K) > 64
    if kind == "normal":
        assert np.all(c[:, 0] >= 0)
    scale = np.abs(dy).max(axis=1, keepdims=True)
    for dxt in (_native.sparsemax_bwd_cols(dyt, cols, p, supp), _native.sparsemax_bwd(dyt, p, supp)):
        assert np.all(np.abs(dxt.cpu().numpy() - dx_ref) <= 1e-6 * scale)


def test_entry_points_capture_into_a_hip_graph(dev):
    """The C-ABI calls allocate nothing and never synchronise, so a forward + list backward pair
    records into a HIP graph (torch.cuda.CUDAGraph is hipGraph on ROCm; the output tensors come
    from the graph's private pool) and replays on new inputs with the same result as eager calls."""
    from lvmhelpers import _native
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(64, 256, device=dev, generator=g)
    dy = torch.randn(64, 256, device=dev, generator=g)
    _native.sparsemax_fwd(x, want_cols=True)            # load the library outside the capture
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            p, supp, _, _, _, cols = _native.sparsemax_fwd(x, want_cols=True)
            dx = _native.sparsemax_bwd_cols(dy, cols, p, supp)
    torch.cuda.current_stream().wait_stream(s)
    for seed in (11, 12):
        x.copy_(torch.randn(64, 256, device=dev, generator=g) * (1 + seed % 2))
        dy.copy_(torch.randn(64, 256, device=dev, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        p_e, supp_e, _, _, _, cols_e = _native.sparsemax_fwd(x, want_cols=True)
        dx_e = _native.sparsemax_bwd_cols(dy, cols_e, p_e, supp_e)
        assert torch.equal(p, p_e) and torch.equal(supp, supp_e) and torch.equal(dx, dx_e)


@pytest.mark.parametrize("K", [1, 2, 31, 32, 33, 127, 128, 129, 256, 257, 512, 513, 1024, 1025, 2047, 2048,
                               2049, 4095, 4097, 8191, 8193, 16384, 16385, 32769, 65535, 65536, 65537])
def test_every_tile_boundary(dev, oracle_mod, K):
    """One shape either side of every switch in the dispatch table (lane-group tiles, one row per
    wave, workgroup tiles, streamed rows; 16-byte and scalar layouts): forward with all optional
    outputs, support lists, both backward entry points, against the oracle."""
    from lvmhelpers import _native
    R = 5
    rng = np.random.default_rng(K)
    x = rng.standard_normal((R, K)).astype(np.float32)
    x[1] *= 0.05
    x[2] = np.round(x[2] * 2) / 2
    dy = rng.standard_normal((R, K)).astype(np.float32)
    p_ref, s_ref, tau_ref, nnz_ref = oracle_mod.sparsemax_fwd(x)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    xt, dyt = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    p, supp, nnz, tau, ent, cols, mask = _native.sparsemax_fwd(
        xt, want_nnz=True, want_tau=True, want_entropy=True, want_cols=True, want_mask=True)
    assert np.array_equal(p.cpu().numpy(), p_ref) and np.array_equal(supp.cpu().numpy(), s_ref)
    assert np.array_equal(tau.cpu().numpy(), tau_ref) and np.array_equal(nnz.cpu().numpy(), nnz_ref)
    np.testing.assert_allclose(ent.cpu().numpy(), oracle_mod.entropy(p_ref), rtol=1e-6, atol=1e-7)
    c = cols.cpu().numpy()
    for r in range(R):
        nz = np.flatnonzero(p_ref[r] > 0)
        if c[r, 0] >= -1:
            assert sorted(c[r, :nz.size].tolist()) == nz.tolist() and np.all(c[r, nz.size:] == -1)
        else:
            assert nz.size > c.shape[1] or K > 4096
    scale = np.maximum(np.abs(dy).max(axis=1, keepdims=True), 1e-30)
    for dxt in (_native.sparsemax_bwd_cols(dyt, cols, p, supp, mask), _native.sparsemax_bwd(dyt, p, supp)):
        assert np.all(np.abs(dxt.cpu().numpy() - dx_ref) <= 1e-6 * scale)


def test_calls_on_two_streams_do_not_interfere(dev):
    """The entry points keep no state between calls: launches queued on two streams at once (each
    with its own tensors) give the results of the same calls made one after the other."""
    from lvmhelpers import _native
    g = torch.Generator(device=dev).manual_seed(8)
    xs = [torch.randn(20000, 1000, device=dev, generator=g) * s for s in (1.0, 0.1)]
    dys = [torch.randn(20000, 1000, device=dev, generator=g) for _ in range(2)]
    ref = []
    for x, dy in zip(xs, dys):
        p, supp, _, _, _, cols = _native.sparsemax_fwd(x, want_cols=True)
        ref.append((p, _native.sparsemax_bwd_cols(dy, cols, p, supp)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    out = [None, None]
    for rep in range(3):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                p, supp, _, _, _, cols = _native.sparsemax_fwd(xs[i], want_cols=True)
                out[i] = (p, _native.sparsemax_bwd_cols(dys[i], cols, p, supp))
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(out[i][0], ref[i][0]) and torch.equal(out[i][1], ref[i][1])


@pytest.mark.parametrize("K", [8, 17, 64, 256, 700, 1500, 4096, 9000, 70000])
def test_fuzz_rows_from_the_algorithm_model(dev, oracle_mod, K):
    """The row generator the exact-edge algorithm was fuzzed with as a numpy model
    (tools/edge_model.py: crowds planted at, one ulp around and just below the oracle's own
    threshold, two-level ties, spikes, rounded rows, several scales) through the kernels: p, supp,
    tau bit-exact, with and without the list / pattern by-products, and dx from all three
    backward forms within 1e-6 of the row scale."""
    from lvmhelpers import _native
    from tools.edge_model import gen_row
    rng = np.random.default_rng(K)
    R = 400 if K <= 4096 else 60
    x = np.stack([gen_row(rng, K) for _ in range(R)]).astype(np.float32)
    p_ref, s_ref, tau_ref, nnz_ref = oracle_mod.sparsemax_fwd(x)
    xt = torch.from_numpy(x).to(dev)
    p, supp, nnz, tau, _ = _native.sparsemax_fwd(xt, want_nnz=True, want_tau=True)
    assert np.array_equal(supp.cpu().numpy(), s_ref)
    assert np.array_equal(p.cpu().numpy(), p_ref)
    assert np.array_equal(tau.cpu().numpy(), tau_ref) and np.array_equal(nnz.cpu().numpy(), nnz_ref)
    p2, supp2, _, _, _, cols, mask = _native.sparsemax_fwd(xt, want_cols=True, want_mask=True)
    assert torch.equal(p2, p) and torch.equal(supp2, supp)
    head = cols[:, 0].cpu().numpy()
    listed = head >= -1
    got = np.sort(np.where(cols.cpu().numpy() >= 0, cols.cpu().numpy(), K + 1), axis=1)
    for r in np.flatnonzero(listed)[:200]:
        assert np.array_equal(got[r][: nnz_ref[r]], np.flatnonzero(p_ref[r] > 0))
    assert np.all(nnz_ref[listed] <= cols.shape[1])
    dy = rng.standard_normal((R, K)).astype(np.float32)
    dyt = torch.from_numpy(dy).to(dev)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    scale = np.abs(dy).max(axis=1, keepdims=True)
    dx_p = _native.sparsemax_bwd(dyt, p, supp)
    assert np.all(np.abs(dx_p.cpu().numpy() - dx_ref) <= 1e-6 * scale)
    for m in (mask, None):
        dx = _native.sparsemax_bwd_cols(dyt, cols, p, supp, m)
        assert np.all(np.abs(dx.cpu().numpy() - dx_ref) <= 1e-6 * scale)


@pytest.mark.parametrize("K,sigma", [(8192, 0.08), (16384, 0.07), (65536, 0.05), (65536, 0.035), (70000, 0.05)])
def test_wide_rows_carry_lists_of_up_to_128_columns(dev, oracle_mod, K, sigma):
    """Rows wider than a wave's tile (K > 4096) get 128-slot lists (lvmhelpers._native.
    support_cols_cap): supports of 33 ... 128 non-zeros, which used to send the backward to the
    p-based gather, are listed and the segment backward reads two slots per lane."""
    from lvmhelpers import _native
    R = 96
    rng = np.random.default_rng(K + int(1000 * sigma))
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    dy = rng.standard_normal((R, K)).astype(np.float32)
    p_ref, s_ref, _, _ = oracle_mod.sparsemax_fwd(x)
    dx_ref = oracle_mod.sparsemax_bwd(dy, p_ref, s_ref)
    p, supp, _, _, _, cols = _native.sparsemax_fwd(torch.from_numpy(x).to(dev), want_cols=True)
    assert cols.shape[1] == 128 == _native.support_cols_cap(K)
    assert np.array_equal(p.cpu().numpy(), p_ref) and np.array_equal(supp.cpu().numpy(), s_ref)
    c = cols.cpu().numpy()
    nnz = (p_ref > 0).sum(1)
    assert ((nnz > 32) & (nnz <= 128)).any()
    long_listed = 0
    for r in range(R):
        if c[r, 0] >= 0:
            nz = np.flatnonzero(p_ref[r] > 0)
            assert nz.size <= 128 and sorted(c[r, :nz.size].tolist()) == nz.tolist()
            assert np.all(c[r, nz.size:] == -1)
            long_listed += nz.size > 64
        else:
            assert c[r, 0] == -2
    assert long_listed > 0 or (nnz <= 64).all() or (nnz > 128).all()
    dx = _native.sparsemax_bwd_cols(torch.from_numpy(dy).to(dev), cols, p, supp).cpu().numpy()
    scale = np.maximum(np.abs(dy).max(axis=1, keepdims=True), 1e-30)
    assert np.all(np.abs(dx - dx_ref) <= 1e-6 * scale)
    # the C ABI refuses longer lists, and 128 for rows of one wave's tile
    import ctypes
    L = _native.lib()
    one = ctypes.c_void_p(p.data_ptr())
    assert L.smlvm_sparsemax_fwd_cols(one, one, one, None, None, None, one, ctypes.c_int32(129), None,
                                      ctypes.c_int64(1), ctypes.c_int32(K), None) == -1
    assert L.smlvm_sparsemax_fwd_cols(one, one, one, None, None, None, one, ctypes.c_int32(128), None,
                                      ctypes.c_int64(1), ctypes.c_int32(4096), None) == -1


@pytest.mark.parametrize("K", [10, 256, 1000, 4096, 8192, 65536, 100000])
def test_masked_logits_minus_inf(dev, oracle_mod, K):
    """Logits masked with -inf (the usual way to exclude classes; entmax's sort / cumsum leaves them
    out of the support): every kernel family gives the sparsemax of the finite entries, zeros at the
    masked ones, and a finite gradient that is zero there -- including a row with one finite entry."""
    from lvmhelpers import functional as F_
    rng = np.random.default_rng(K)
    R = 8
    x = rng.standard_normal((R, K)).astype(np.float32)
    m = rng.random((R, K)) < 0.5
    m[:, 0] = False
    x[m] = -np.inf
    x[1, 1:] = -np.inf
    p_ref, _, _, _ = oracle_mod.sparsemax_fwd(x)
    for r in range(R):
        f = ~np.isinf(x[r])
        sub, _, _, _ = oracle_mod.sparsemax_fwd(x[r:r + 1, f])
        assert np.array_equal(sub[0], p_ref[r, f]) and np.all(p_ref[r, ~f] == 0)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    p = F_.sparsemax(xt, dim=-1)
    dy = torch.randn(R, K, device=dev)
    (p * dy).sum().backward()
    assert np.array_equal(p.detach().cpu().numpy(), p_ref)
    g = xt.grad.cpu().numpy()
    assert np.isfinite(g).all() and np.all(g[np.isinf(x)] == 0)
    assert p_ref[1, 0] == 1.0


@pytest.mark.parametrize("K", [10, 256, 4096, 8192, 65536])
def test_masked_logits_minus_inf_entmax15(dev, oracle_mod, K):
    """The same for entmax15 (floating point: 2e-6 on p against the oracle on the finite entries)."""
    from lvmhelpers import functional as F_
    rng = np.random.default_rng(K + 1)
    R = 6
    x = rng.standard_normal((R, K)).astype(np.float32)
    m = rng.random((R, K)) < 0.5
    m[:, 0] = False
    x[m] = -np.inf
    x[1, 1:] = -np.inf
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    p = F_.entmax15(xt, dim=-1)
    (p * torch.randn(R, K, device=dev)).sum().backward()
    pn = p.detach().cpu().numpy()
    for r in range(R):
        f = ~np.isinf(x[r])
        sub = oracle_mod.entmax15_fwd(x[r:r + 1, f])[0]
        assert np.abs(pn[r, f] - sub[0]).max() <= 2e-6 and np.all(pn[r, ~f] == 0)
    g = xt.grad.cpu().numpy()
    assert np.isfinite(g).all() and np.all(g[np.isinf(x)] == 0)
