"""CPU model of the exact threshold search of csrc/sparsemax.hip (solve_core / solve_stream), used to
fuzz the ALGORITHM against oracle/sparsemax_oracle.c before it is written as a kernel:

    python tools/edge_model.py [seconds]

It follows the kernel choice by choice (Michelot start, robust-pass / robust-fail classification of
tie groups by a rounding-noise bound, exact rank counts inside groups the bound cannot decide, the
candidate floor that makes elements outside the list provably irrelevant) but runs serially in
numpy; it is development tooling, not product code and not the oracle.
"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import oracle  # noqa: E402

F = np.float32
EPS23 = 2.0 ** -23


def cond_rank(m, S, v, r):
    j = F(m + r)
    Sx = S + float(r) * float(v)
    c = F(F(Sx) - F(1.0))
    return bool(F(j * F(v)) > c)


def edge(z, lo):
    gt = z > lo
    lt = z < lo
    return dict(m=int(gt.sum()), t=int((z == lo).sum()), S=float(z[gt].astype(np.float64).sum()),
                vnext=float(z[lt].max()) if lt.any() else -np.inf)


def classify(m, S, v, t):
    d = float(m) * float(v) - S + 1.0
    B = EPS23 * (3.0 * (m + t) * abs(float(v)) + 1.0)
    return 1 if d > B else (-1 if d < -B else 0)


def count_exact(m, S, v, t):
    return sum(1 for r in range(1, t + 1) if cond_rank(m, S, v, r))


def cand_floor(K):
    """Elements <= floor can neither be in the support nor pass the oracle's fp32 test by rounding."""
    a = 3.0 * K * EPS23
    u = (1.0 + EPS23) / (1.0 - a)
    return F(-(u * (1.0 + 1e-6)))


def lazy_list_is_complete(thr, tau_k, m_robust, K):
    a = F(3.0) * F(K) * F(EPS23)
    m = F(m_robust)
    tl = F(abs(F(tau_k)) + F(1e-5))
    return bool(m > a and F(abs(F(thr)) * F(m - a)) > F(F(m * tl + F(EPS23)) * F(1.0 + 1e-5)))


def solve(zv, nvis, stats):
    """zv: visible shifted values (max = 0).  Returns (s, S, m_top)."""
    z = zv.astype(F)
    tau = F(-1.0)
    cp = 1 << 60
    while True:
        inn = z > tau
        c = int(inn.sum())
        if c >= cp:
            break
        cp = c
        s = F(z[inn].sum(dtype=F))
        tau = max(tau, F((s - F(1.0)) / F(c)))
    # first pass at tau: the set {z > tau}, its smallest value and the largest value outside it; the
    # tie group at lo has f = 1 + m_all lo - S_all (its own t lo cancels)
    inn = z > tau
    lo = F(z[inn].min())
    m_top, S_top = int(inn.sum()), float(z[inn].astype(np.float64).sum())
    vnext = float(z[~inn].max()) if (~inn).any() else -np.inf
    stats["edge"] += 1
    if not (float(m_top) * float(lo) - S_top + 1.0 > EPS23 * (3.0 * m_top * abs(float(lo)) + 1.0)):
        while True:
            b = edge(z, lo)
            stats["edge"] += 1
            if b["m"] == 0 or classify(b["m"], b["S"], lo, b["t"]) > 0:
                break
            lo = F(z[z > lo].min())
            stats["up"] += 1
        m_top = b["m"] + b["t"]
        S_top = b["S"] + float(b["t"]) * float(lo)
        vnext = b["vnext"]
    m, S = m_top, S_top
    first_lo = vnext
    C, opened, dirty, S_at = 0, True, False, S_top
    while vnext > -np.inf:
        d = float(m) * vnext - S + 1.0
        if d < -EPS23 * (3.0 * nvis * abs(vnext) + 1.0):
            break
        g = edge(z, F(vnext))
        stats["edge"] += 1
        stats["down"] += 1
        k = classify(g["m"], g["S"], F(vnext), g["t"])
        cnt = g["t"] if k > 0 else (0 if k < 0 else count_exact(g["m"], g["S"], F(vnext), g["t"]))
        if k == 0:
            stats["count"] += 1
        if opened:
            S_at = g["S"] + float(cnt) * vnext
            opened = cnt == g["t"]
        elif cnt > 0:
            dirty = True
        C += cnt
        m = g["m"] + g["t"]
        S = g["S"] + float(g["t"]) * vnext
        vnext = g["vnext"]
    s = m_top + C
    if dirty:
        stats["dirty"] += 1
        rem, lo2 = C, first_lo
        while True:
            g = edge(z, F(lo2))
            take = min(rem, g["t"])
            S_at = g["S"] + float(take) * lo2
            rem -= take
            if rem == 0:
                break
            lo2 = g["vnext"]
    return s, S_at, m_top


def model_row(x, use_floor=True, stats=None, cap=256):
    """use_floor: candidate path (list of the elements above the floor; when more than `cap`, the
    lazy list above a Michelot lower bound - 1e-3 with the a-posteriori completeness check)."""
    stats = stats if stats is not None else dict(edge=0, up=0, down=0, count=0, dirty=0)
    stats.setdefault("lazy", 0)
    stats.setdefault("lazy_rejected", 0)
    K = x.shape[0]
    z = (x - x.max()).astype(F)
    vis = z
    if use_floor:
        floor = cand_floor(K)
        vis = z[z > floor]
        if vis.shape[0] > cap:
            # Michelot steps on the whole row until few enough remain
            tau, cp, vis = F(-1.0), 1 << 60, z
            for _ in range(9):
                inn = z > tau
                c = int(inn.sum())
                if c >= cp:
                    break
                cp = c
                tau = max(tau, F((F(z[inn].sum(dtype=F)) - F(1.0)) / F(c)))
                if c <= 2 * cap:
                    thr = max(F(tau - F(1e-3)), floor)
                    lst = z[z > thr]
                    if lst.shape[0] <= cap:
                        stats["lazy"] += 1
                        s, S, m_top = solve(lst, cap, stats)
                        if thr <= floor or lazy_list_is_complete(thr, tau, m_top, K):
                            vis = None
                        else:
                            stats["lazy_rejected"] += 1
                        break
            if vis is None:
                tau_f = F(F(F(S) - F(1.0)) / F(s))
                return np.maximum(z - tau_f, F(0)).astype(F), s, tau_f
    s, S, _ = solve(vis, vis.shape[0], stats)
    tau = F(F(F(S) - F(1.0)) / F(s))
    p = np.maximum(z - tau, F(0)).astype(F)
    return p, s, tau


def gen_row(rng, K):
    kind = rng.integers(0, 12)
    sig = [1.0, 0.3, 0.05, 0.01, 3.0][rng.integers(0, 5)]
    x = (sig * rng.standard_normal(K)).astype(F)
    if kind == 0:
        x[:] = F(rng.standard_normal())
    elif kind == 1:
        x[:] = -1.0
        x[rng.integers(0, K)] = F(rng.choice([0.0, 0.002, 1e-5, -1e-5, 3.0, 0.02]))
    elif kind == 2:
        x = (np.round(x * 4) / 4).astype(F)
    elif kind in (3, 4, 5, 6, 7):
        # plant crowds at / around the oracle's own threshold
        _, s0, tau0, _ = oracle.sparsemax_fwd(x[None])
        t = F(tau0[0] + x.max())
        below = np.flatnonzero(x < t)
        rng.shuffle(below)
        n = int(rng.integers(1, max(2, below.size)))
        idx = below[:n]
        if kind == 3:
            x[idx] = t
        elif kind == 4:
            x[idx] = np.nextafter(t, F(-np.inf))
        elif kind == 5:
            x[idx] = np.nextafter(t, F(np.inf))
        elif kind == 6:
            h = n // 3
            x[idx[:h]] = t
            x[idx[h:2 * h]] = np.nextafter(t, F(-np.inf))
            x[idx[2 * h:]] = np.nextafter(t, F(np.inf))
        else:
            x[idx] = t - F(rng.choice([1e-5, 1e-6, 1e-4, 2e-3, 3e-7]))
    elif kind == 8:
        x[: K // 2] = 1.0
    elif kind == 9:
        # one spike, everything else tied just around spike - 1
        x[:] = 0.0
        x[0] = F(1.0 + rng.choice([0.0, 1e-6, -1e-6, 2e-3, -2e-3, 1e-4, 2e-2]))
    elif kind == 10:
        # two-level ties: a values at 0, b at -delta
        a = int(rng.integers(1, max(2, K // 2)))
        x[:] = F(-rng.choice([1.0 / a, 1.0 / a + 1e-6, 1.0 / a - 1e-6, 0.5 / a]))
        x[:a] = 0.0
    return x


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0 = time.time()
    n = bad = 0
    stats = dict(edge=0, up=0, down=0, count=0, dirty=0)
    while time.time() - t0 < budget:
        K = int(rng.choice([3, 8, 17, 64, 256, 700, 1500, 4096, 9000, 70000],
                           p=[.1, .1, .1, .15, .15, .12, .1, .1, .05, .03]))
        x = gen_row(rng, K)
        p_ref, s_ref, tau_ref, _ = oracle.sparsemax_fwd(x[None])
        p, s, tau = model_row(x, use_floor=bool(rng.integers(0, 2)), stats=stats,
                              cap=int(rng.choice([8, 32, 256])))
        n += 1
        if s != s_ref[0] or tau != tau_ref[0] or not np.array_equal(p, p_ref[0]):
            bad += 1
            if bad <= 10:
                print("MISMATCH K", K, "s", s, s_ref[0], "tau", tau, tau_ref[0])
                np.save(f"/tmp/edge_bad_{bad}.npy", x)
    print(f"{n} rows, {bad} mismatches, stats {stats}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
