"""CPU: the binary sequence factor of the SparseMAP oracle (sequence_smap, lvmhelpers/sparsemap.py:
47-87,130-148; sequence.h / sequence_binary.h) against brute force.  lpsmap is absent (parity
unpinned by the reference); what pins the restatement: the linear oracle against enumeration, the
QP's optimal value and unique marginals against an independent solver over all 2^n
configurations, the optimality gap, and the Jacobian against finite differences."""
import itertools

import numpy as np
import pytest


def all_configs(n):
    return np.array(list(itertools.product([0, 1], repeat=n)), np.float64)


def cfg_scores(Y, x, t):
    return Y @ x + (Y[:, :-1] * Y[:, 1:]) @ t


def full(C):
    return np.concatenate([C, 1 - C], axis=1)


def qp_solve(Y, s, iters=400000):
    """min_p 0.5 |M p|^2 - s.p over the simplex, columns of M = [y; 1 - y] (the factor's 2n binary
    variables: M^T M counts the positions where two configurations agree, sequence.h:198-213);
    FISTA on p with exact simplex projection.  Returns (marginals of the on-variables, objective)."""
    def proj(v):
        u = np.sort(v)[::-1]
        css = np.cumsum(u) - 1
        k = np.nonzero(u - css / np.arange(1, len(u) + 1) > 0)[0][-1]
        return np.maximum(v - css[k] / (k + 1), 0)
    M = full(Y)
    G = M @ M.T
    Lc = np.linalg.eigvalsh(G)[-1]
    p = np.full(len(Y), 1.0 / len(Y))
    q, tk = p.copy(), 1.0
    for _ in range(iters):
        pn = proj(q - (G @ q - s) / Lc)
        tn = (1 + np.sqrt(1 + 4 * tk * tk)) / 2
        q = pn + (tk - 1) / tn * (pn - p)
        done = np.abs(pn - p).max() < 1e-15
        p, tk = pn, tn
        if done:
            break
    return Y.T @ p, 0.5 * p @ G @ p - s @ p


@pytest.mark.parametrize("n", [2, 3, 5, 8])
def test_viterbi_is_the_argmax(oracle_mod, n):
    """With an iteration limit of 0 the only vertex is Maximize(eta, t)."""
    rng = np.random.default_rng(n)
    Y = all_configs(n)
    for _ in range(30):
        x = rng.standard_normal(n).astype(np.float32)
        t = (2 * rng.standard_normal(n - 1)).astype(np.float32)
        out = oracle_mod.sparsemap_seq_fwd(x[None], t[None], None, max_iter=0)
        y = oracle_mod.unpack_bits(out["aset_bits"], n)[0, 0].astype(np.float64)
        sc = cfg_scores(Y, x.astype(np.float64), t.astype(np.float64))
        assert abs(cfg_scores(y[None], x.astype(np.float64), t.astype(np.float64))[0] - sc.max()) < 1e-12


@pytest.mark.parametrize("n,scale", [(2, 1.0), (3, 2.0), (4, 1.0), (6, 0.5), (8, 1.0), (8, 3.0), (10, 1.0)])
def test_solution_matches_brute_force_qp(oracle_mod, n, scale):
    rng = np.random.default_rng(10 * n + int(scale))
    Y = all_configs(n)
    B = 16
    x = (scale * rng.standard_normal((B, n))).astype(np.float32)
    t = (scale * rng.standard_normal((B, n - 1))).astype(np.float32)
    init = rng.random((B, 2 * n))
    for ini in (None, init):
        out = oracle_mod.sparsemap_seq_fwd(x, t, ini, max_iter=300)
        assert (out["status"] == 0).all(), out["status"]
        cfgs = oracle_mod.unpack_bits(out["aset_bits"], n).astype(np.float64)
        for b in range(B):
            m = out["alen"][b]
            p = out["p64"][b, :m]
            C = cfgs[b, :m]
            assert abs(p.sum() - 1) < 1e-10 and (p > -1e-12).all()
            xb, tb = x[b].astype(np.float64), t[b].astype(np.float64)
            s = cfg_scores(Y, xb, tb)
            mu = C.T @ p
            mu2 = np.concatenate([mu, 1 - mu])
            obj = 0.5 * mu2 @ mu2 - cfg_scores(C, xb, tb) @ p
            if n <= 4:   # an independent first-order solver converges quickly on 16 configurations
                mu_ref, obj_ref = qp_solve(Y, s, 50000)
                assert obj <= obj_ref + 1e-8              # at least as good as its optimum
                assert np.abs(mu - mu_ref).max() < 1e-4   # the marginals are unique
            # the optimality certificate of a convex QP over the simplex, on ALL 2^n configurations:
            # every partial derivative is at least the one shared by the vertices that carry weight
            grad = full(Y) @ mu2 - s
            gact = full(C) @ mu2 - cfg_scores(C, xb, tb)
            carried = gact[p > 1e-9]
            assert carried.max() - carried.min() < 1e-7
            assert grad.min() >= carried.min() - 1e-7


def test_larger_chains_converge_and_are_optimal(oracle_mod):
    """n = 32 and 128: no brute force; every sample converges and passes the oracle's own
    optimality test evaluated independently (Viterbi by dynamic programming in numpy)."""
    def viterbi_value(on, off, t):
        v0, v1 = off[0], on[0]
        for i in range(1, len(on)):
            v0, v1 = max(v0, v1) + off[i], max(v0, v1 + t[i - 1]) + on[i]
        return max(v0, v1)
    rng = np.random.default_rng(0)
    for n, B in ((32, 24), (128, 8)):
        x = rng.standard_normal((B, n)).astype(np.float32)
        t = rng.standard_normal((B, n - 1)).astype(np.float32)
        out = oracle_mod.sparsemap_seq_fwd(x, t, rng.random((B, 2 * n)), max_iter=1000)
        # nearly dependent vertices (the new one a hair outside the hull of the active ones) make
        # the active-set method zigzag: a few samples run into the iteration limit, as upstream's would
        assert (out["status"] <= 1).all() and (out["status"] == 0).mean() >= 0.9, (out["status"], out["iters"])
        cfgs = oracle_mod.unpack_bits(out["aset_bits"], n).astype(np.float64)
        for b in np.flatnonzero(out["status"] == 0):
            m = out["alen"][b]
            p = out["p64"][b, :m]
            C = cfgs[b, :m]
            assert abs(p.sum() - 1) < 1e-9 and (p > -1e-12).all()
            xb, tb = x[b].astype(np.float64), t[b].astype(np.float64)
            mu = C.T @ p
            # max over ALL configurations of score - <M_y, u> (by Viterbi) equals the active ones' value
            best = viterbi_value(xb - mu, -(1 - mu), tb)
            act = cfg_scores(C, xb, tb) - full(C) @ np.concatenate([mu, 1 - mu])
            assert best <= act[p > 1e-9].max() + 1e-6


def test_jacobian_by_finite_differences(oracle_mod):
    n, B = 6, 8
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, n)).astype(np.float32)
    t = rng.standard_normal((B, n - 1)).astype(np.float32)
    out = oracle_mod.sparsemap_seq_fwd(x, t, None, max_iter=300)
    cap = n + 1
    dp = rng.standard_normal((B, cap)).astype(np.float32)
    ds, dt = oracle_mod.sparsemap_seq_bwd(dp, out["aset_bits"], out["alen"], out["invA"], n)
    eps = 1e-3

    def loss(xx, tt, b):
        o = oracle_mod.sparsemap_seq_fwd(xx[None], tt[None], None, max_iter=300)
        if o["alen"][0] != out["alen"][b]:
            return None
        # same vertices (any order): distribution dotted with dp in the reference order
        mine = {tuple(r): i for i, r in enumerate(o["aset_bits"][0, :o["alen"][0]].tolist())}
        acc = 0.0
        for i, r in enumerate(out["aset_bits"][b, :out["alen"][b]].tolist()):
            if tuple(r) not in mine:
                return None
            acc += float(o["p64"][0, mine[tuple(r)]]) * float(dp[b, i])
        return acc

    checked = 0
    for b in range(B):
        for k in range(n):
            xp, xm = x[b].copy(), x[b].copy()
            xp[k] += eps
            xm[k] -= eps
            lp, lm = loss(xp, t[b], b), loss(xm, t[b], b)
            if lp is None or lm is None:
                continue
            assert abs((lp - lm) / (2 * eps) - ds[b, k]) < 2e-3
            checked += 1
        for k in range(n - 1):
            tp, tm = t[b].copy(), t[b].copy()
            tp[k] += eps
            tm[k] -= eps
            lp, lm = loss(x[b], tp, b), loss(x[b], tm, b)
            if lp is None or lm is None:
                continue
            assert abs((lp - lm) / (2 * eps) - dt[b, k]) < 2e-3
            checked += 1
    assert checked > 20
