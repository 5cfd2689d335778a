#!/usr/bin/env python3
"""Benchmark of the sparse-marginalization hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic logits: sparsemax forward +
backward over a [rows, K] fp32 matrix resident in HBM (BASELINE.json configs[2], the synthetic
sweep the metric is quoted on; default 1M rows x K=4096 per GPU, N(0,1) logits).  For N > 1 the
driver launches this file with torch.distributed.run; rows are sharded per rank with no data-path
collective (weak scaling: every rank owns `rows` rows), the timed region is bracketed by a barrier
+ synchronize on both sides and the maximum over ranks is reported.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel: algorithmic bytes per launch / mean launch time (HIP events on the
                launch stream inside the timed region) against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/, plain C, 1 thread) timed on this host on a bounded sample
  cpu_baseline_torch_all_cores  entmax's sort/cumsum algorithm in torch CPU ops on all host threads
  sweep         fwd+bwd rows/s at the other K of the sweep (256, 65536)
  compaction    support count / compaction / weighted reduction kernels vs their algorithmic bytes
  sparsemap     batched active-set solver: samples/s and active-set iterations/s (N > 1: every rank
                solves 1024 samples at 128 bits = BASELINE.json configs[4] sharded, whole-job rate)
  train_steps   one optimisation step of each BASELINE.json model config on synthetic data (N > 1:
                config 5 with the flat-bucket gradient all-reduce over RCCL)
  sparsemap_cpu_baseline  the fp64 C oracle of the solver timed on one host core, same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sparse-marginalization-lvm_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6300


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1 << 20, help="rows per GPU")
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--sigma", type=float, default=1.0, help="logits ~ N(0, sigma^2)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the K=256/65536 and SparseMAP extras")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    return ap.parse_args()


def dist_setup(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("SMLVM_BENCH_FORCE_DIST") == "1":
        # (the env switch lets a 1-GPU box rehearse the RCCL code path with a single rank)
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return world, rank, local


def _dist_on():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def barrier(world):
    if _dist_on():
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if not _dist_on():
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_sparsemax(native, x, dy, steps, warmup, world, dev):
    """Returns (wall seconds for `steps` fwd+bwd passes, mean fwd ms, mean bwd ms)."""
    R, K = x.shape
    p = torch.empty_like(x)
    dx = torch.empty_like(x)
    supp = torch.empty(R, dtype=torch.int32, device=dev)
    cap = native.SUPPORT_COLS_CAP
    cols = torch.empty((R, cap), dtype=torch.int32, device=dev)
    L = native.lib()
    import ctypes
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    mbytes = L.smlvm_sparsemax_support_mask_bytes(ctypes.c_int64(R), ctypes.c_int32(K))
    mask = torch.empty(mbytes // 4, dtype=torch.int32, device=dev) if mbytes else None
    mp = ctypes.c_void_p(mask.data_ptr()) if mask is not None else None

    # the pair lvmhelpers.functional.sparsemax runs: the forward also lists each row's support
    # columns (128 B per row) and the backward works from the lists (include/sparsemarg.h)
    def fwd():
        rc = L.smlvm_sparsemax_fwd_cols(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(p.data_ptr()),
                                        ctypes.c_void_p(supp.data_ptr()), None, None, None,
                                        ctypes.c_void_p(cols.data_ptr()), ctypes.c_int32(cap), mp,
                                        ctypes.c_int64(R), ctypes.c_int32(K), st)
        assert rc == 0, rc

    def bwd():
        rc = L.smlvm_sparsemax_bwd_cols(ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(cols.data_ptr()),
                                        ctypes.c_int32(cap), mp, ctypes.c_void_p(p.data_ptr()),
                                        ctypes.c_void_p(supp.data_ptr()), ctypes.c_void_p(dx.data_ptr()),
                                        ctypes.c_int64(R), ctypes.c_int32(K), st)
        assert rc == 0, rc

    for _ in range(warmup):
        fwd(); bwd()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    barrier(world)
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record(); fwd(); ev[i][1].record(); bwd(); ev[i][2].record()
    barrier(world)
    wall = time.perf_counter() - t0
    f_all = sorted(e[0].elapsed_time(e[1]) for e in ev)
    b_all = sorted(e[1].elapsed_time(e[2]) for e in ev)
    f_ms, b_ms = sum(f_all) / steps, sum(b_all) / steps
    timed_sparsemax.last_median = (f_all[steps // 2], b_all[steps // 2])
    mean_supp = float(supp.float().mean().item())
    del p, dx
    return wall, f_ms, b_ms, mean_supp


def gen(R, K, sigma, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.empty(R, K, device=dev, dtype=torch.float32)
    x.normal_(0.0, sigma, generator=g)
    dy = torch.empty(R, K, device=dev, dtype=torch.float32)
    dy.normal_(0.0, 1.0, generator=g)
    return x, dy


def cpu_baseline(K, sigma):
    """The plain-C oracle (1 thread) on a bounded sample of the same workload, ~10 s."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(0)

    def run(rows):
        x = (sigma * rng.standard_normal((rows, K))).astype(np.float32)
        dy = rng.standard_normal((rows, K)).astype(np.float32)
        t0 = time.perf_counter()
        p, s, _, _ = oracle.sparsemax_fwd(x)
        oracle.sparsemax_bwd(dy, p, s)
        return time.perf_counter() - t0

    probe_rows = max(8, (1 << 22) // K)
    t = run(probe_rows)
    rows = int(max(probe_rows, min(10.0 / max(t / probe_rows, 1e-9), (1 << 28) // K)))
    t = run(rows)
    return {"value": rows / t, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"{rows} rows x K={K} fp32 N(0,{sigma}^2), oracle/sparsemax_oracle.c fwd+bwd, "
                      f"{t:.1f} s on 1 of {os.cpu_count()} host cores"}


def cpu_baseline_torch(K, sigma):
    """entmax's algorithm in torch CPU ops (what the reference runs on a CPU), all host threads."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(1)
    rows = max(64, (1 << 25) // K)          # 128 MB of logits
    x = (sigma * rng.standard_normal((rows, K))).astype(np.float32)
    dy = rng.standard_normal((rows, K)).astype(np.float32)
    oracle.torch_sparsemax_fwd_bwd(x[:64], dy[:64])
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 5.0:
        oracle.torch_sparsemax_fwd_bwd(x, dy)
        reps += 1
    t = time.perf_counter() - t0
    return {"value": rows * reps / t, "unit": "rows/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} x {rows} rows x K={K} fp32 N(0,{sigma}^2), entmax's sort/cumsum sparsemax fwd+bwd "
                      f"in torch CPU ops, {t:.1f} s on {torch.get_num_threads()} threads of {os.cpu_count()} host cores"}


def sparsemap_cpu_baseline(B, n, budget, sigma):
    """The fp64 C oracle (1 thread) on a bounded sample of the cfg5 SparseMAP workload."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(77)
    x = (sigma * rng.standard_normal((B, n))).astype(np.float32)
    init = torch.rand(B, 2 * n, dtype=torch.double, generator=torch.Generator().manual_seed(42)).numpy()
    t0 = time.perf_counter()
    ref = oracle.sparsemap_fwd(x, init, budget, 300)
    t = time.perf_counter() - t0
    return {"value": B / t, "unit": "samples/s", "cores": 1, "kind": "port",
            "active_set_iters_per_s": float(ref["iters"].sum()) / t,
            "sample": f"{B} samples x {n} bits, budget {budget}, N(0,{sigma}^2) scores, "
                      f"oracle/sparsemap_oracle.c forward, {t:.1f} s on 1 of {os.cpu_count()} host cores"}


def topk_bench(native, dev, B, n, k, reps=20, cpu=True):
    """k-best bit vectors of N(0,1) scores, one launch for the batch; next to it
    the reference's own binary_topk.h compiled in place (oracle/_ref, 1 thread) when that build
    travelled with the tree, else our C restatement of it."""
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, n, device=dev, generator=g)
    native.binary_topk_batched(x, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        native.binary_topk_batched(x, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out = {"B": B, "bits": n, "k": k, "ms": ms, "samples_per_s": B / (ms * 1e-3),
           "note": "includes the allocation of the [B,k,n] fp32 output by the Python wrapper"}
    if cpu:
        import oracle
        xs = x[:min(B, 2048)].cpu().numpy()
        kind, fn = "port", oracle.binary_topk_batched
        if oracle.ref_lib() is not None:
            kind, fn = "reference", oracle.ref_binary_topk_batched
        t0 = time.perf_counter()
        fn(xs, k)
        t = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": xs.shape[0] / t, "unit": "samples/s", "cores": 1, "kind": kind,
                               "sample": f"{xs.shape[0]} samples, {t * 1e3:.1f} ms on 1 host core"}
    return out


def sparsemap_bench(native, dev, B, n, budget, sigma, reps=5, world=1, seed=77):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, n, device=dev, generator=g) * sigma
    init = torch.rand(B, 2 * n, dtype=torch.double,
                      generator=torch.Generator().manual_seed(42)).to(dev)
    out = native.sparsemap_fwd(x, init, budget, 300)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = native.sparsemap_fwd(x, init, budget, 300)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    dp = torch.randn(B, n + 1, device=dev, generator=g)
    e0.record()
    for _ in range(reps):
        native.sparsemap_bwd(dp, out["aset_bits"], out["alen"], out["inv_a"], n)
    e1.record()
    torch.cuda.synchronize()
    bms = e0.elapsed_time(e1) / reps
    iters = float(out["iters"].double().sum().item())
    res = {"B": B, "bits": n, "budget": budget, "sigma": sigma, "fwd_ms": ms, "bwd_ms": bms,
           "samples_per_s": B / (ms * 1e-3), "active_set_iters_per_s": iters / (ms * 1e-3),
           "mean_iters": iters / B, "mean_support": float(out["alen"].float().mean().item()),
           "converged_frac": float((out["status"] != 1).float().mean().item())}
    if _dist_on():  # whole-job figures: every rank solves its own B samples, slowest rank counts
        ms_max = max_over_ranks(ms, world, dev)
        res.update({"n_gpus": world, "B_total": B * world, "fwd_ms_max_over_ranks": ms_max,
                    "samples_per_s": B * world / (ms_max * 1e-3)})
    return res


def compaction_bench(native, dev, R, K, sigma, reps=5):
    """Kernel 2 on the probabilities of a forward pass: support_count, support_offsets,
    support_compact and the weighted loss reduction, each against its algorithmic HBM bytes
    (count 4KR read; compact 4KR + 12N + 8(R+1); reduce 12N read)."""
    x, _ = gen(R, K, sigma, dev, 4321)
    p, supp, nnz, _, _, lists = native.sparsemax_fwd(x, want_nnz=True, want_cols=True)
    del x

    def timeit(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_count = timeit(lambda: native.support_count(p))
    t_off = timeit(lambda: native.support_offsets(nnz))
    off, rows, cols, vals = native.support_compact(p, nnz)
    N = int(rows.shape[0])
    t_all = timeit(lambda: native.support_compact(p, nnz))       # offsets + one sync + compact
    comps = torch.rand(N, device=dev) * 500
    t_red = timeit(lambda: native.weighted_loss_reduce_fwd(vals, comps, off, 1.0 / R))
    t_compact = max(t_all - t_off, 1e-6)
    gb = lambda b, ms: b / (ms * 1e-3) / 1e9
    # the same from the forward's support lists: count + offsets + sync + compact, no pass over p
    t_lists = timeit(lambda: native.support_compact(p, support_cols=lists))
    fused_bytes = 12.0 * N + 8.0 * (R + 1) + 2 * 4.0 * lists.shape[1] * R
    return {"R": R, "K": K, "N": N,
            "from_support_lists": {"ms_count_offsets_compact": t_lists,
                                   "ms_scan_count_offsets_compact": t_count + t_all,
                                   "algorithmic_bytes": fused_bytes,
                                   "note": "what the Marginalizer / top-k paths run: rows listed by the "
                                           "forward are counted and compacted from <= cap entries"},
            "support_count": {"ms": t_count, "GBps": gb(4.0 * K * R, t_count)},
            "support_offsets": {"ms": t_off},
            "support_compact": {"ms": t_compact, "GBps": gb(4.0 * K * R + 12.0 * N + 8.0 * (R + 1), t_compact),
                                "note": "offsets launch subtracted; includes the one host sync that sizes the lists"},
            "weighted_loss_reduce_fwd": {"ms": t_red, "GBps": gb(12.0 * N, t_red)},
            "frac_of_hbm_peak_count": gb(4.0 * K * R, t_count) / HBM_PEAK_GBS,
            "frac_of_hbm_peak_compact": gb(4.0 * K * R + 12.0 * N + 8.0 * (R + 1), t_compact) / HBM_PEAK_GBS}


def train_steps_bench(dev, world, rank):
    """One optimisation step (forward, backward, gradient all-reduce when N > 1, Adam) of each
    BASELINE.json model config on synthetic data of the experiment's shape (bench_models.py)."""
    import bench_models as bm

    def run(model, batch, B, steps=5, warm=2):
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        out = None
        for _ in range(warm):
            out = bm.train_step(model, opt, batch, distributed=_dist_on())
        barrier(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = bm.train_step(model, opt, batch, distributed=_dist_on())
        barrier(world)
        ms = max_over_ranks((time.perf_counter() - t0) / steps * 1e3, world, dev)
        sup = out["log"].get("support")
        return {"ms_per_step": ms, "samples_per_s": B * world / (ms * 1e-3), "batch_per_gpu": B,
                "mean_support": float(torch.as_tensor(sup, dtype=torch.float32).mean()) if sup is not None else None,
                "loss": float(out["loss"])}

    res = {}
    if not _dist_on():
        m, b = bm.make_ssvae(dev, 64)
        res["cfg1_ssvae_marg_sparsemax_K10_b64"] = run(m, b, 64)
        m, b = bm.make_ssvae(dev, 64, compact=True)
        res["cfg1_ssvae_marg_sparsemax_K10_b64_compact"] = run(m, b, 64)
        m, b = bm.make_signal_game(dev, 64)
        res["cfg2_signal_game_marg_sparsemax_K256_b64"] = run(m, b, 64)
        m, b = bm.make_signal_game(dev, 64, compact=True)
        res["cfg2_signal_game_marg_sparsemax_K256_b64_compact"] = run(m, b, 64)
        m, b = bm.make_bitvae(dev, 64, 32, "sparsemap")
        res["cfg4_bitvae_sparsemap_32bits_b64"] = run(m, b, 64)
        m, b = bm.make_bitvae(dev, 64, 32, "sparsemap", fused_decoder=False)
        res["cfg4_bitvae_sparsemap_32bits_b64_dense_logits"] = run(m, b, 64)
        m, b = bm.make_bitvae(dev, 64, 32, "topksparse")
        res["cfg4_bitvae_topksparse_32bits_b64"] = run(m, b, 64)
        del m, b
    # cfg5: global batch 8192 over 8 GPUs = 1024 per rank, one flat-bucket RCCL all-reduce per step
    m, b = bm.make_bitvae(dev, 1024, 128, "sparsemap", seed=0)
    res["cfg5_bitvae_sparsemap_128bits_b1024_per_gpu"] = run(m, b, 1024, steps=3, warm=1)
    res["cfg5_bitvae_sparsemap_128bits_b1024_per_gpu"]["grad_allreduce"] = (
        f"1 bucket of {sum(p.numel() for p in m.parameters()) * 4 / 1e6:.1f} MB over {world} rank(s)")
    return res


def main():
    a = parse()
    world, rank, local = dist_setup(a.gpus)
    dev = torch.device("cuda", local if _dist_on() else 0)
    from lvmhelpers import _native as native
    native.lib()

    x, dy = gen(a.rows, a.K, a.sigma, dev, 1234 + rank)
    wall, f_ms, b_ms, mean_supp = timed_sparsemax(native, x, dy, a.steps, a.warmup, world, dev)
    wall = max_over_ranks(wall, world, dev)
    del x, dy
    torch.cuda.empty_cache()

    rows_total = a.rows * world * a.steps
    fwd_bytes = 8.0 * a.K * a.rows
    bwd_bytes = 12.0 * a.K * a.rows
    f_med, b_med = timed_sparsemax.last_median
    kern = {
        "sparsemax_fwd": {"ms": f_ms, "ms_median": f_med, "algorithmic_bytes": fwd_bytes,
                          "GBps": fwd_bytes / (f_ms * 1e-3) / 1e9},
        "sparsemax_bwd": {"ms": b_ms, "ms_median": b_med, "algorithmic_bytes": bwd_bytes,
                          "GBps": bwd_bytes / (b_ms * 1e-3) / 1e9},
    }
    dom = "sparsemax_fwd" if f_ms >= b_ms else "sparsemax_bwd"
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile) and a.sigma == 1.0:   # counters were collected on the default workload
        try:
            tj = json.load(open(tfile))
            traffic = tj.get(f"{dom}_K{a.K}_R{a.rows}")
            for k in kern:
                kern[k]["hbm_traffic_bytes"] = tj.get(f"{k}_K{a.K}_R{a.rows}")
                if kern[k]["hbm_traffic_bytes"]:   # rate of the bytes that really crossed the HBM pins
                    kern[k]["hbm_GBps"] = kern[k]["hbm_traffic_bytes"] / (kern[k]["ms"] * 1e-3) / 1e9
        except Exception:
            traffic = None
    out = {
        "metric": "sparsemax_fwd_bwd_rows_per_sec",
        "value": rows_total / wall,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"synthetic sparsemax fwd+bwd sweep, {a.rows} rows/GPU x K={a.K} fp32, "
                               f"logits N(0,{a.sigma}^2) (BASELINE.json configs[2])",
                   "rows_per_gpu": a.rows, "K": a.K, "mean_support": mean_supp,
                   "parallelism": f"rows sharded over {world} GPU(s), no collective"},
        "fwd_bwd_GBps_per_gpu": (fwd_bytes + bwd_bytes) / ((f_ms + b_ms) * 1e-3) / 1e9,
        "fwd_bwd_frac_of_hbm_peak": (fwd_bytes + bwd_bytes) / ((f_ms + b_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["GBps"],
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["GBps"] / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "note": "achieved = algorithmic bytes (fwd 8K, bwd 12K per row) / measured time. The "
                             "forward also writes each row's support list (<= 32 columns); the backward "
                             "reads that list, gathers dy at those columns and writes dx without reading p "
                             "or streaming dy, so its HBM traffic (PMC, profiles/traffic.json) is about a "
                             "third of its algorithmic bytes; see `kernels` for both launches (`hbm_GBps` "
                             "= PMC traffic / time)."},
        "kernels": kern,
    }

    if rank == 0 and not _dist_on() and not a.no_sweep:
        sweep = {}

        def adversarial(R, K):
            """SURVEY.md 8(d) S1 adversarial rows, a quarter each: all equal (support K), one
            dominant logit (support 1), an exact tie at the threshold, plain N(0,1)."""
            xs, dys = gen(R, K, 1.0, dev, 99)
            kind = torch.arange(R, device=dev) % 4
            xs[kind == 0] = 0.25
            xs[kind == 1] = -1.0
            xs[kind == 1, 7 % K] = 3.0
            xs[kind == 2] = -1.0
            xs[kind == 2, 0] = 0.0
            return xs, dys

        def sweep_entry(tag, K, R, chunks, make):
            xs, dys = make()
            w, fm, bm, ms_ = timed_sparsemax(native, xs, dys, max(3, a.steps // 4), 2, 1, dev)
            del xs, dys
            torch.cuda.empty_cache()
            gb = 20.0 * K * R / ((fm + bm) * 1e-3) / 1e9
            sweep[tag] = {"rows_per_launch": R, "launches_per_1M_rows": chunks, "fwd_ms": fm,
                          "bwd_ms": bm, "rows_per_s": R / ((fm + bm) * 1e-3), "GBps": gb,
                          "frac_of_hbm_peak": gb / HBM_PEAK_GBS, "mean_support": ms_}

        # K in {256, 4096, 65536} x {N(0,1), N(0,0.05^2), adversarial}; the
        # headline line above is (4096, N(0,1)).  GBps is against the algorithmic 20 K bytes/row.
        for K, R, chunks in ((256, 1 << 20, 1), (4096, 1 << 20, 1), (65536, 1 << 15, 32)):
            if K != a.K or a.sigma != 1.0:
                sweep_entry(f"K{K}", K, R, chunks, lambda: gen(R, K, 1.0, dev, 99))
            sweep_entry(f"K{K}_sigma0.05", K, R, chunks, lambda: gen(R, K, 0.05, dev, 99))
            sweep_entry(f"K{K}_adversarial", K, R, chunks, lambda: adversarial(R, K))
        sweep_entry("K4096_sigma0.2", 4096, 1 << 20, 1, lambda: gen(1 << 20, 4096, 0.2, dev, 99))
        out["sweep"] = sweep
        # the model-shaped calls (cfg1 [64,10], cfg2 [64,256], cfg4 top-k [64,10]):
        # far too small for bandwidth, reported as launch latency of the fwd+bwd pair
        small = {}
        for R, K in ((64, 10), (64, 256)):
            xs, dys = gen(R, K, 1.0, dev, 5)
            w, fm, bm, _ = timed_sparsemax(native, xs, dys, 200, 20, 1, dev)
            small[f"{R}x{K}"] = {"fwd_us": fm * 1e3, "bwd_us": bm * 1e3, "pair_wall_us": w / 200 * 1e6}
            # the same pair recorded once into a HIP graph (the C ABI allocates and synchronises
            # nothing) and replayed: what a launch-bound inner loop would pay
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    pp, ss, _, _, _, cc = native.sparsemax_fwd(xs, want_cols=True)
                    native.sparsemax_bwd_cols(dys, cc, pp, ss)
            torch.cuda.current_stream().wait_stream(side)
            graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                graph.replay()
            torch.cuda.synchronize()
            small[f"{R}x{K}"]["pair_graph_replay_us"] = (time.perf_counter() - t0) / 200 * 1e6
            del graph
        out["model_shaped_latency"] = small
        out["compaction"] = [compaction_bench(native, dev, 1 << 20, 4096, a.sigma),
                             compaction_bench(native, dev, 1 << 20, 256, a.sigma),
                             compaction_bench(native, dev, 1 << 20, 256, 0.05)]
        out["binary_topk"] = [topk_bench(native, dev, 64, 32, 10, cpu=not a.no_cpu),
                              topk_bench(native, dev, 64, 128, 10, cpu=not a.no_cpu),
                              topk_bench(native, dev, 8192, 128, 10, cpu=not a.no_cpu)]
        out["sparsemap"] = [sparsemap_bench(native, dev, 64, 32, 0, 1.0),
                            sparsemap_bench(native, dev, 64, 32, 16, 1.0),
                            sparsemap_bench(native, dev, 1024, 128, 0, 1.0),
                            sparsemap_bench(native, dev, 1024, 128, 64, 1.0),
                            sparsemap_bench(native, dev, 1024, 128, 0, 3.0),
                            sparsemap_bench(native, dev, 8192, 128, 0, 1.0)]
    if _dist_on() and not a.no_sweep:
        # cfg5: batch 8192 over 8 GPUs = 1024 samples per rank at 128 bits; samples sharded, no collective
        out["sparsemap"] = [sparsemap_bench(native, dev, 1024, 128, 0, 1.0, world=world, seed=77 + rank),
                            sparsemap_bench(native, dev, 1024, 128, 0, 3.0, world=world, seed=77 + rank)]
    if not a.no_sweep:
        torch.cuda.empty_cache()
        ts = train_steps_bench(dev, world, rank)   # every rank takes part (collective inside)
        out["train_steps"] = ts
    if rank == 0 and not _dist_on() and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(a.K, a.sigma)
        out["cpu_baseline_torch_all_cores"] = cpu_baseline_torch(a.K, a.sigma)
        out["sparsemap_cpu_baseline"] = sparsemap_cpu_baseline(1024, 128, 0, 1.0)
    if rank == 0:
        print(json.dumps(out))
    if _dist_on():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
