"""CPU: pins the sparsemax oracle (oracle/sparsemax_oracle.c).

The reference arithmetic lives in the absent third-party `entmax`, so the oracle is pinned by
(a) a torch-CPU restatement of entmax's sort/cumsum algorithm (bit-exact: proves the fp64-accumulated
prefix sums are what torch's CPU cumsum does), (b) an independent bisection solver, (c) closed
forms and invariants (SURVEY.md 8(c) i-vi)."""
import numpy as np
import pytest
import torch

import oracle


def torch_sort_sparsemax(x):
    """entmax's published algorithm with torch CPU ops (max-shift, sort, cumsum, strict test)."""
    X = torch.from_numpy(x)
    X = X - X.max(dim=-1, keepdim=True)[0]
    srt, _ = torch.sort(X, dim=-1, descending=True)
    cs = srt.cumsum(-1) - 1
    rho = torch.arange(1, X.shape[-1] + 1, dtype=X.dtype)
    support = rho * srt > cs
    s = support.sum(-1, keepdim=True)
    tau = cs.gather(-1, s - 1) / s.to(X.dtype)
    return torch.clamp(X - tau, min=0).numpy(), s.squeeze(-1).numpy().astype(np.int32), tau.squeeze(-1).numpy()


def bisect_sparsemax(x):
    """fp64 bisection on tau followed by the closed form on the found support."""
    z = x.astype(np.float64) - x.max(-1, keepdims=True)
    lo, hi = np.full(z.shape[0], -1.0), np.zeros(z.shape[0])
    for _ in range(60):
        mid = (lo + hi) / 2
        g = np.maximum(z - mid[:, None], 0).sum(-1) - 1
        lo, hi = np.where(g > 0, mid, lo), np.where(g > 0, hi, mid)
    supp = z > hi[:, None]
    tau = (np.where(supp, z, 0).sum(-1) - 1) / supp.sum(-1)
    return np.maximum(z - tau[:, None], 0), supp


@pytest.mark.parametrize("R,K,sigma", [(64, 10, 1.0), (64, 256, 1.0), (32, 4096, 1.0),
                                       (4, 65536, 1.0), (64, 256, 0.05), (8, 4096, 0.05),
                                       (64, 37, 10.0), (200, 1, 1.0), (200, 2, 1.0)])
def test_matches_torch_cpu_semantics_bit_exact(R, K, sigma):
    rng = np.random.default_rng(K)
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    p, s, tau, nnz = oracle.sparsemax_fwd(x)
    p_t, s_t, tau_t = torch_sort_sparsemax(x)
    assert np.array_equal(s, s_t)
    assert np.array_equal(tau, tau_t)
    assert np.array_equal(p, p_t)
    assert np.array_equal(nnz, (p > 0).sum(-1))


def test_fp64_prefix_sums_like_torch_cumsum():
    # a row where an fp32 running sum would differ from the fp64-accumulated one
    x = np.full((1, 3000), -1e-4, np.float32)
    x[0, 0] = 0.0
    x[0, 1:40] -= np.float32(3e-8) * np.arange(39, dtype=np.float32)
    p, s, tau, _ = oracle.sparsemax_fwd(x)
    p_t, s_t, tau_t = torch_sort_sparsemax(x)
    assert np.array_equal(s, s_t) and np.array_equal(tau, tau_t) and np.array_equal(p, p_t)


@pytest.mark.parametrize("K,sigma", [(10, 1.0), (256, 1.0), (256, 0.05), (4096, 1.0), (4096, 0.02)])
def test_invariants_and_independent_solver(K, sigma):
    rng = np.random.default_rng(K + 1)
    x = (sigma * rng.standard_normal((50, K))).astype(np.float32)
    p, s, tau, nnz = oracle.sparsemax_fwd(x)
    assert np.all(p >= 0)
    assert np.all(np.abs(p.astype(np.float64).sum(-1) - 1) <= 1.2e-7 * K + 1e-6)          # (i)
    z = x - x.max(-1, keepdims=True)
    assert np.array_equal(p > 0, z > tau[:, None])                                          # (ii)
    p_b, supp_b = bisect_sparsemax(x)
    margin = np.abs(z.astype(np.float64) - (z.astype(np.float64) - p_b).max(-1, keepdims=True))
    decided = np.abs(p_b - 0) > 1e-6
    assert np.array_equal((p > 0)[decided | ~supp_b & (margin > 1e-6)],
                          supp_b[decided | ~supp_b & (margin > 1e-6)])                      # (iii)
    assert np.abs(p - p_b).max() < 2e-6
    p_shift, s_shift, _, _ = oracle.sparsemax_fwd(x + np.float32(2.0))                      # (vi)
    assert np.abs(p_shift - p).max() < 2e-6


def test_closed_forms():
    K = 64
    p, s, _, _ = oracle.sparsemax_fwd(np.full((1, K), 0.3, np.float32))                      # (iv)
    assert np.array_equal(s, [K]) and np.allclose(p, 1.0 / K, rtol=1e-6)
    x = np.zeros((1, K), np.float32); x[0, 5] = 1.0
    p, s, _, _ = oracle.sparsemax_fwd(x)
    assert s[0] == 1 and p[0, 5] == 1.0 and p.sum() == 1.0
    x2 = np.array([[0.3, -0.1], [2.0, -2.0], [0.0, 0.0]], np.float32)
    p, _, _, _ = oracle.sparsemax_fwd(x2)
    assert np.allclose(p[:, 0], np.clip((x2[:, 0] - x2[:, 1] + 1) / 2, 0, 1), atol=1e-7)
    # exact tie at the threshold stays out of the support (strict test)
    x3 = np.full((1, 8), -1.0, np.float32); x3[0, 0] = 0.0
    p, s, tau, nnz = oracle.sparsemax_fwd(x3)
    assert s[0] == 1 and nnz[0] == 1 and tau[0] == -1.0


def test_backward_vs_finite_differences():                                                  # (v)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((20, 16)).astype(np.float32)
    dy = rng.standard_normal((20, 16)).astype(np.float32)
    p, s, tau, _ = oracle.sparsemax_fwd(x)
    dx = oracle.sparsemax_bwd(dy, p, s)
    z = x - x.max(-1, keepdims=True)
    safe = np.abs(z - tau[:, None]).min(-1) > 1e-2       # rows away from a support change
    h = 1e-3
    for r in np.nonzero(safe)[0][:8]:
        for j in range(16):
            xp = x[r:r + 1].astype(np.float64).copy(); xm = xp.copy()
            xp[0, j] += h; xm[0, j] -= h
            fp = (bisect_sparsemax(xp)[0] * dy[r]).sum(); fm = (bisect_sparsemax(xm)[0] * dy[r]).sum()
            assert abs((fp - fm) / (2 * h) - dx[r, j]) < 1e-4
    assert np.array_equal(dx == 0, p == 0) or np.all(dx[p == 0] == 0)


def test_entropy_and_loss():
    rng = np.random.default_rng(9)
    x = rng.standard_normal((32, 10)).astype(np.float32)
    p, _, _, _ = oracle.sparsemax_fwd(x)
    eps = np.finfo(np.float32).eps
    pc = np.clip(p, eps, 1 - eps)
    ref = -np.where(p > 0, pc * np.log(pc), 0).sum(-1)
    assert np.allclose(oracle.entropy(p), ref, rtol=1e-6, atol=1e-7)
    onehot = np.zeros((1, 5), np.float32); onehot[0, 2] = 1
    assert abs(oracle.entropy(onehot)[0] - 1.19e-7) < 2e-8      # not 0: SURVEY.md 8(a) a2
    t = rng.integers(0, 10, 32)
    loss, grad = oracle.sparsemax_loss(x, t)
    ref_loss = (1 - (p.astype(np.float64) ** 2).sum(-1)) / 2 + (p * x).sum(-1) - x[np.arange(32), t]
    assert np.allclose(loss, ref_loss, rtol=1e-5, atol=1e-6)
    assert np.all(loss >= -1e-6)
    e = np.zeros_like(p); e[np.arange(32), t] = 1
    assert np.allclose(grad, p - e)


def test_compaction_order_is_torch_nonzero_order():
    rng = np.random.default_rng(1)
    p = rng.random((40, 33)).astype(np.float32)
    p[rng.random(p.shape) < 0.8] = 0
    off, rows, cols, vals = oracle.support_compact(p)
    nz = torch.nonzero(torch.from_numpy(p) > 0).numpy()
    assert np.array_equal(rows, nz[:, 0]) and np.array_equal(cols, nz[:, 1])
    assert np.array_equal(vals, p[p > 0])
    assert np.array_equal(off, np.concatenate([[0], np.cumsum((p > 0).sum(-1))]))


def test_torch_all_cores_baseline_matches_c_oracle(oracle_mod):
    """bench.py's all-cores CPU baseline (oracle.torch_sparsemax_fwd_bwd) computes the same thing
    as the C restatement: p, supp bit-exact; dx within 1e-6 of the row scale."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((257, 300)).astype(np.float32)
    dy = rng.standard_normal((257, 300)).astype(np.float32)
    p, s, _, _ = oracle_mod.sparsemax_fwd(x)
    dx = oracle_mod.sparsemax_bwd(dy, p, s)
    p2, s2, dx2 = oracle_mod.torch_sparsemax_fwd_bwd(x, dy)
    assert np.array_equal(p, p2) and np.array_equal(s, s2)
    assert np.all(np.abs(dx - dx2) <= 1e-6 * np.abs(dy).max(axis=1, keepdims=True))
