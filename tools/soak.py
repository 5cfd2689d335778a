"""GPU box: randomized parity soak with FRESH seeds (the test suite's seeds are fixed): sparsemax
forward (p, supp bit-exact), list backward (against the p-based backward and the oracle), the
compaction lists, the SparseMAP solver and the sequence factor (everything identical to the oracle).
    python tools/soak.py [seconds] [seed] [other]     (other: k-best bit vectors and entmax15 instead)
Prints one line per 50 trials and every mismatch with what reproduces it; exit status 1 on any."""
import sys
import time
import numpy as np
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
import oracle
from lvmhelpers import _native

dev = torch.device('cuda:0')
budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
print(f"soak: {budget_s:.0f} s, base seed {seed0}", flush=True)
bad = 0


def rows(rng, R, K):
    kind = int(rng.integers(0, 9))
    sigma = float(rng.choice([0.003, 0.02, 0.05, 0.2, 1.0, 3.0, 30.0]))
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    if kind == 0:
        x = np.round(x * 4) / 4                       # many exact ties
    elif kind == 1:
        x[:] = x[:, :1]                               # all equal
    elif kind == 2:
        x[:] = -1.0
        x[np.arange(R), rng.integers(0, K, R)] = 0.0  # K - 1 elements exactly at the threshold
    elif kind == 3:
        x += np.float32(rng.choice([-1e4, 1e4, 3e7]))  # large common offset (fp32 spacing of ties)
    elif kind == 4:
        x[:, : max(1, K // 3)] = x[:, :1]             # a third of the row tied
    elif kind == 5:
        x = np.sort(x, axis=1)                        # sorted rows (lists fill from one end)
    return x, (kind, sigma)


def mixed_rows(rng, R, K):
    """Rows of several kinds side by side in random order (rows of K <= 512 share a wave, and the
    few-valued shortcuts of round 3 decide per row inside it), plus two-valued rows: a zeros over a
    plateau at v."""
    parts, tags = [], []
    left = R
    while left > 0:
        r = int(rng.integers(1, left + 1))
        if rng.random() < 0.35:
            v = float(rng.choice([-1.0, -0.5, -0.999, -1.0 - 2 ** -20, -1.0 + 2 ** -20, -1e-3, -2.0, -1.0 / 3]))
            base = float(rng.choice([0.0, 0.25, -3.0, 100.0]))
            x = np.full((r, K), np.float32(base + v), np.float32)
            for i in range(r):
                a = int(rng.integers(1, max(2, min(K, int(rng.choice([2, 5, 40, K]))))))
                x[i, rng.choice(K, min(a, K), replace=False)] = np.float32(base)
            tag = ("two-valued", v)
        else:
            x, tag = rows(rng, r, K)
        parts.append(x)
        tags.append(tag)
        left -= r
    x = np.concatenate(parts)[rng.permutation(R)]
    return x, ("mixed", tuple(tags[:4]))


def sparsemax_trial(rng, trial):
    global bad
    K = int(rng.choice([rng.integers(1, 64), rng.integers(64, 1100), rng.integers(1100, 4200),
                        rng.integers(4200, 70000), rng.choice([256, 1024, 4096, 8192, 65536, 131072])]))
    R = int(rng.integers(1, max(2, min(300, (1 << 21) // K))))
    x, tag = mixed_rows(rng, R, K) if rng.random() < 0.5 else rows(rng, R, K)
    p_ref, s_ref, _, _ = oracle.sparsemax_fwd(x)
    xt = torch.from_numpy(x).to(dev)
    out = _native.sparsemax_fwd(xt, want_cols=True, want_mask=True)
    p, supp, cols, mask = out[0], out[1], out[5], out[6]
    if not (np.array_equal(p.cpu().numpy(), p_ref) and np.array_equal(supp.cpu().numpy(), s_ref)):
        bad += 1
        print(f"MISMATCH sparsemax fwd: trial {trial} R {R} K {K} kind/sigma {tag}", flush=True)
        return
    dy = rng.standard_normal((R, K)).astype(np.float32)
    dyt = torch.from_numpy(dy).to(dev)
    dx_cols = _native.sparsemax_bwd_cols(dyt, cols, p, supp, mask).cpu().numpy()
    dx_ref = oracle.sparsemax_bwd(dy, p_ref, s_ref)
    scale = np.abs(dy).max()
    if not np.allclose(dx_cols, dx_ref, atol=2e-6 * scale, rtol=0):
        bad += 1
        print(f"MISMATCH sparsemax bwd: trial {trial} R {R} K {K} kind/sigma {tag} "
              f"max err {np.abs(dx_cols - dx_ref).max():.3e}", flush=True)
    # compaction fed by the lists == compaction by scanning p == oracle
    off_ref, r_ref, c_ref, v_ref = oracle.support_compact(p_ref)
    for lists in (None, cols):
        off, rws, cls, vals = _native.support_compact(p, want_vals=True, support_cols=lists)
        if not (np.array_equal(rws.cpu().numpy(), r_ref) and np.array_equal(cls.cpu().numpy(), c_ref)
                and np.array_equal(vals.cpu().numpy(), v_ref) and np.array_equal(off.cpu().numpy(), off_ref)):
            bad += 1
            print(f"MISMATCH compaction ({'lists' if lists is not None else 'scan'}): trial {trial} R {R} K {K} "
                  f"kind/sigma {tag}", flush=True)


def sparsemap_trial(rng, trial):
    global bad
    seq = rng.random() < 0.25
    n = int(rng.integers(2 if seq else 1, 129))
    B = int(rng.integers(1, 200))
    sigma = float(rng.choice([0.1, 0.2, 0.5, 1.0, 2.0, 4.0, 8.0]))
    budget = 0 if (seq or rng.random() < 0.4) else int(rng.integers(1, n + 1))
    max_iter = int(rng.choice([1, 2, 5, 17, 100, 300, 1000]))
    use_init = bool(rng.random() < 0.7)
    x = (sigma * rng.standard_normal((B, n))).astype(np.float32)
    q = rng.random()
    if q < 0.25:
        x = np.round(x)
    elif q < 0.35:
        x = np.round(x * 2) / 2
    init = rng.random((B, 2 * n)) if use_init else None
    it = torch.from_numpy(init).to(dev) if use_init else None
    if seq:
        t = (sigma * rng.standard_normal((B, n - 1))).astype(np.float32)
        ref = oracle.sparsemap_seq_fwd(x, t, init, max_iter=max_iter)
        out = _native.sparsemap_seq_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev), it, max_iter)
    else:
        ref = oracle.sparsemap_fwd(x, init, budget, max_iter)
        out = _native.sparsemap_fwd(torch.from_numpy(x).to(dev), it, budget, max_iter)
    ok = all(np.array_equal(out[k].cpu().numpy(), ref[k]) for k in ("status", "iters", "alen", "p"))
    ok = ok and np.array_equal(out["aset_bits"].cpu().numpy().view(np.uint32), ref["aset_bits"])
    if not ok:
        bad += 1
        print(f"MISMATCH sparsemap: trial {trial} seq {seq} n {n} B {B} sigma {sigma} budget {budget} "
              f"max_iter {max_iter} init {use_init} q {q:.2f}", flush=True)


def topk_trial(rng, trial):
    global bad
    n = int(rng.integers(1, 200))
    k = int(rng.integers(2, 65)) if rng.random() < 0.7 else int(rng.integers(65, 1025))
    if n < 11:
        k = min(k, 1 << n)
    if k < 2:
        return
    B = int(rng.integers(1, 300))
    sc = (float(rng.choice([0.1, 1.0, 5.0])) * rng.standard_normal((B, n))).astype(np.float32)
    q = rng.random()
    if q < 0.3:
        sc = np.round(sc * 2) / 2
    elif q < 0.4:
        sc[:] = 0.0
    cfg_ref, top_ref = oracle.binary_topk_batched(sc, k)
    cfg, top = _native.binary_topk_batched(torch.from_numpy(sc).to(dev), k, want_scores=True)
    if not (np.array_equal(top.cpu().numpy(), top_ref) and np.array_equal(cfg.cpu().numpy(), cfg_ref)):
        bad += 1
        print(f"MISMATCH topk: trial {trial} B {B} n {n} k {k} q {q:.2f}", flush=True)


def entmax15_trial(rng, trial):
    """floating point: p to 2e-6, dx to 2e-5 of the row scale (tests/test_gpu_entmax15.py)"""
    global bad
    K = int(rng.choice([rng.integers(1, 64), rng.integers(64, 1100), rng.integers(1100, 4200),
                        rng.integers(4200, 65537)]))
    R = int(rng.integers(1, max(2, min(200, (1 << 20) // K))))
    x, tag = rows(rng, R, K)
    if tag[0] == 3:
        return   # offsets of 1e4 ... 3e7: x / 2 - max is exact but the oracle's fp32 sort/cumsum is not the bar there
    p_ref = oracle.entmax15_fwd(x)[0]
    p = _native.entmax15_fwd(torch.from_numpy(x).to(dev))[0]
    dy = rng.standard_normal((R, K)).astype(np.float32)
    dx = _native.entmax15_bwd(torch.from_numpy(dy).to(dev), p).cpu().numpy()
    dx_ref = oracle.entmax15_bwd(dy, p.cpu().numpy())
    e_p = np.abs(p.cpu().numpy() - p_ref).max()
    e_dx = np.abs(dx - dx_ref).max() / max(1e-30, np.abs(dy).max())
    if not (e_p <= 2e-6 and e_dx <= 2e-5):
        bad += 1
        print(f"MISMATCH entmax15: trial {trial} R {R} K {K} kind/sigma {tag} p err {e_p:.2e} dx err {e_dx:.2e}",
              flush=True)


only = sys.argv[3] if len(sys.argv) > 3 else ""
t0 = time.time()
trial = 0
while time.time() - t0 < budget_s:
    rng = np.random.default_rng([seed0, trial])
    if only == "other":
        (topk_trial if trial % 2 else entmax15_trial)(rng, trial)
    elif trial % 3 == 2:
        sparsemap_trial(rng, trial)
    else:
        sparsemax_trial(rng, trial)
    trial += 1
    if trial % 50 == 0:
        print(f"  {trial} trials, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"soak done: {trial} trials, {bad} mismatches (base seed {seed0})", flush=True)
sys.exit(1 if bad else 0)
