#!/usr/bin/env python3
"""Benchmark of the sparse-marginalization hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic logits: sparsemax forward +
backward over a [rows, K] fp32 matrix resident in HBM (BASELINE.json configs[2], the synthetic
sweep the metric is quoted on; default 1M rows x K=4096 per GPU, N(0,1) logits).

`--gpus N` with N > 1, started as a plain `python bench.py --gpus N ...`, is the PARENT of the N
ranks: it touches no GPU, starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port <free port> bench.py <same flags>` as a child, relays the
child's JSON line and exits with its status.  A rank (RANK in the environment) binds cuda:LOCAL_RANK,
joins the RCCL group and refuses to run if WORLD_SIZE differs from --gpus.  Rows are sharded per
rank with no data-path collective (weak scaling: every rank owns `rows` rows); the timed region is
bracketed by a barrier + synchronize on both sides and the maximum over ranks is reported; rank 0
prints ONE JSON line (`ranks_seen`, `rank_checksums_p_dx` and `rccl_version` show that N ranks ran).

Extra objects on that line:
  roofline      dominant kernel (forward): SURVEY.md 8(d)'s algorithmic bytes per launch (8 K per
                row) / mean launch time (HIP events on the launch stream inside the timed region)
                against the 8 TB/s HBM peak; `traffic` = HBM bytes per launch from profiles/traffic.json
  kernels, fwd_bwd_frac   per launch: the least bytes it has to move (min_traffic below: forward
                8K + supp + list, list backward 4K + list + one 64-byte sector per support element,
                ...) / time / 8 TB/s -- no fraction on the line can exceed 1
  cpu_baseline  the CPU oracle (oracle/, plain C, 1 thread) timed on this host on a bounded sample
  cpu_baseline_torch_all_cores  entmax's sort/cumsum algorithm in torch CPU ops on all host threads
  sweep         SURVEY.md 8(d) S1: K in {256, 4096, 65536} x {N(0,1), N(0,0.05^2), adversarial
                rows}, 20 launches each, medians
  compaction    support count / offsets / compaction / weighted reduction kernels through the C ABI
                on preallocated buffers (median of 20) vs the bytes each has to move
  sparsemap     batched active-set solver, SURVEY.md 8(d) S5 grid: samples/s, active-set
                iterations/s, converged / stalled fractions (N > 1: every rank solves 1024 samples
                at 128 bits = BASELINE.json configs[4] sharded, whole-job rate)
  sparsemap_sequence  the binary sequence factor (sequence_smap) on the same solver, 2 shapes
  train_steps   one optimisation step of each BASELINE.json model config on synthetic data (N > 1:
                config 5 with the flat-bucket gradient all-reduce over RCCL); config 2 also in the
                support-only and hoisted-decoder modes
  sparsemap_cpu_baseline  the fp64 C oracle of the solver timed on one host core, same workload
  sparsemap_cpu_baseline_all_cores  the same on every host thread of this process
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


def child_command(n_gpus, argv, port):
    """The command the parent starts for `--gpus N` when it was not itself started by a launcher:
    one rank per GPU of this node over RCCL (the driver's own multi-GPU command line)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as a
    CHILD process (this parent never touches the GPU, and never replaces itself), relay rank 0's
    JSON line and leave with the child's exit code -- a failed rank makes the whole run fail."""
    import subprocess
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} asked for, {have} visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(child_command(a.gpus, argv, _free_port()), env=env, stdout=subprocess.PIPE,
                          text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {a.gpus}-rank run failed (exit {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    if json.loads(line).get("n_gpus") != a.gpus:
        print("bench.py: the child reported a different rank count", file=sys.stderr)
        return 1
    print(line)
    return 0


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


def gather_max(K):
    """Support size up to which the p-based backward gathers dy instead of streaming it
    (csrc/sparsemax.hip gather_max / kGatherMaxSupport)."""
    return min(K >> 4, 96)


def min_traffic(K, cap, cols, supp, has_mask):
    """Least HBM bytes one launch has to move, per kernel, given what the rows turned out to be:
      forward        read x, write p, supp and the support list: 8K + 4 + 4 cap per row
                     (+ 512 B for a row that leaves a support mask)
      backward       row with a list:  write dx, read list + supp, one 64 B sector of dy per listed
                                       column: 4K + 4 cap + 4 + 64 s
                     row with a mask:  4K + 512 + 64 + 4 + 64 s  (1024 < K <= 4096, s <= 96;
                                       8K + 512 + 64 + 4 beyond: dy streamed);
                                       8K + 2G + 64 + 4 for 32 < K <= 1024 (dy streamed, G lanes per row)
                     no list, s <= gather_max: read p, write dx, gather dy: 8K + 4 + 64 s
                     otherwise: stream dy and p, write dx: 12K + 4
    These are the denominators of every `frac` below (never the 20 K model, which the list backward
    undercuts by not reading p or dy at all)."""
    R = supp.shape[0]
    head = cols[:, 0]
    s = supp.to(torch.float64)
    listed = head >= -1
    small = 32 < K <= 1024    # 16-register tiles: every row leaves a pattern of 2 G bytes
    pat = 2.0 * (8 if K <= 128 else 16 if K <= 256 else 32 if K <= 512 else 64)
    if not has_mask:
        masked = torch.zeros_like(listed)
    elif small:
        masked = head == -3
    else:
        masked = head == -3
    wide = masked & (supp > gather_max(K)) & (not small)   # pattern rows that stream dy: 8K + 512
    masked = masked & ~wide
    gath = ~listed & ~masked & ~wide & (supp <= gather_max(K))
    stream = ~listed & ~masked & ~wide & ~gath
    n_masked_fwd = int((head == -3).sum().item()) if has_mask else 0
    fwd = R * (8.0 * K + 4 + 4 * cap) + (pat * R if (small and has_mask) else 512.0 * n_masked_fwd)
    # a pattern row of a 16-register tile streams dy: 8K + pattern + list head + supp
    masked_row = (8.0 * K + pat + 64 + 4) if small else (4.0 * K + 512 + 64 + 4)
    masked_sectors = 0.0 if small else 64.0 * float(s[masked].sum().item())
    bwd = (float(listed.sum().item()) * (4.0 * K + 4 * cap + 4) + 64.0 * float(s[listed].sum().item())
           + float(masked.sum().item()) * masked_row + masked_sectors
           + float(wide.sum().item()) * (8.0 * K + 512 + 64 + 4)
           + float(gath.sum().item()) * (8.0 * K + 4) + 64.0 * float(s[gath].sum().item())
           + float(stream.sum().item()) * (12.0 * K + 4))
    return {"fwd": fwd, "bwd": bwd,
            "rows": {"listed": int(listed.sum().item()), "masked": int(masked.sum().item()) + int(wide.sum().item()),
                     "gather": int(gath.sum().item()), "stream": int(stream.sum().item())}}


def checksum64(t):
    """Order-independent 8-byte checksum of a tensor's bits (wrapping int64 sum of its 32-bit words)."""
    return int(t.view(torch.int32).sum(dtype=torch.int64).item())


def timed_sparsemax(native, x, dy, steps, warmup, world, dev):
    """`steps` fwd+bwd passes.  Returns a dict: wall seconds, mean/median fwd and bwd ms (HIP events
    on the launch stream), mean support, the least-traffic model of both launches and checksums."""
    R, K = x.shape
    p = torch.empty_like(x)
    dx = torch.empty_like(x)
    supp = torch.empty(R, dtype=torch.int32, device=dev)
    cap = native.support_cols_cap(K)
    cols = torch.empty((R, cap), dtype=torch.int32, device=dev)
    L = native.lib()
    import ctypes
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    mbytes = L.smlvm_sparsemax_support_mask_bytes(ctypes.c_int64(R), ctypes.c_int32(K))
    mask = torch.zeros(mbytes // 4, dtype=torch.int32, device=dev) if mbytes else None
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
    res = {"wall": wall, "fwd_ms": sum(f_all) / steps, "bwd_ms": sum(b_all) / steps,
           "fwd_ms_median": f_all[steps // 2], "bwd_ms_median": b_all[steps // 2],
           "launches": steps, "mean_support": float(supp.float().mean().item()),
           "min_traffic": min_traffic(K, cap, cols, supp, mask is not None),
           "checksum_p": checksum64(p), "checksum_dx": checksum64(dx)}
    del p, dx
    return res


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


def sparsemap_cpu_baseline_all_cores(B, n, budget, sigma):
    """the all-cores variant -- the same C oracle on every host thread this process
    may use (ctypes releases the GIL; the batch is split into one slice per thread)."""
    import numpy as np
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(torch.get_num_threads(), os.cpu_count() or 1))
    rng = np.random.default_rng(78)
    x = (sigma * rng.standard_normal((B, n))).astype(np.float32)
    init = torch.rand(B, 2 * n, dtype=torch.double, generator=torch.Generator().manual_seed(43)).numpy()
    cuts = np.linspace(0, B, threads + 1).astype(int)
    oracle.lib()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(lambda i: oracle.sparsemap_fwd(np.ascontiguousarray(x[cuts[i]:cuts[i + 1]]),
                                                         np.ascontiguousarray(init[cuts[i]:cuts[i + 1]]),
                                                         budget, 300), range(threads)))
    t = time.perf_counter() - t0
    iters = float(sum(r["iters"].sum() for r in res))
    return {"value": B / t, "unit": "samples/s", "cores": threads, "kind": "port",
            "active_set_iters_per_s": iters / t,
            "sample": f"{B} samples x {n} bits, budget {budget}, N(0,{sigma}^2) scores, "
                      f"oracle/sparsemap_oracle.c forward on {threads} threads of {os.cpu_count()} host cores, {t:.1f} s"}


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


def median_ms(fn, launches=20, warm=5):
    """SURVEY.md 8(d) timing method: HIP-event pairs around each of >= 20 launches after 5 warm-ups,
    median (events on torch's current stream, which is the stream every call here launches on)."""
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(launches)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return t[launches // 2]


def marginal_error(out, x, n, budget):
    """Per sample: max_k |(M p)_k - mu*_k| against the closed-form marginals (Bernoulli:
    clip((x+1)/2, 0, 1); budget: its projection onto sum mu <= budget, bisection on the multiplier),
    from the fp32 distribution the kernel returns -- EVERY sample, whatever its exit status."""
    k = torch.arange(n, device=x.device)
    mu = torch.zeros(x.shape[0], n, dtype=torch.float64, device=x.device)
    for c0 in range(0, out["p"].shape[1], 16):     # chunks of the active set: bounded temporaries
        bits = ((out["aset_bits"][:, c0:c0 + 16, k // 32] >> (k % 32)) & 1).double()
        mu += torch.einsum("bc,bcn->bn", out["p"][:, c0:c0 + 16].double(), bits)
    a = (x.double() + 1) / 2
    ref = a.clamp(0, 1)
    if budget > 0:
        lo = torch.zeros(x.shape[0], 1, dtype=torch.float64, device=x.device)
        hi = a.max(-1, keepdim=True)[0].clamp(min=0)
        for _ in range(100):
            mid = (lo + hi) / 2
            over = (a - mid).clamp(0, 1).sum(-1, keepdim=True) > budget
            lo, hi = torch.where(over, mid, lo), torch.where(over, hi, mid)
        ref = torch.where(ref.sum(-1, keepdim=True) <= budget, ref, (a - hi).clamp(0, 1))
    return (mu - ref).abs().max(dim=-1)[0]


def sparsemap_bench(native, dev, B, n, budget, sigma, reps=20, world=1, seed=77):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, n, device=dev, generator=g) * sigma
    init = torch.rand(B, 2 * n, dtype=torch.double,
                      generator=torch.Generator().manual_seed(42)).to(dev)
    out = native.sparsemap_fwd(x, init, budget, 300)
    ms = median_ms(lambda: native.sparsemap_fwd(x, init, budget, 300), reps, 2)
    dp = torch.randn(B, n + 1, device=dev, generator=g)
    bms = median_ms(lambda: native.sparsemap_bwd(dp, out["aset_bits"], out["alen"], out["inv_a"], n), reps, 2)
    iters = float(out["iters"].double().sum().item())
    ok = out["status"] == 0
    err = marginal_error(out, x, n, budget)
    res = {"B": B, "bits": n, "budget": budget, "sigma": sigma, "fwd_ms": ms, "bwd_ms": bms,
           "launches": reps, "samples_per_s": B / (ms * 1e-3), "active_set_iters_per_s": iters / (ms * 1e-3),
           "mean_iters": iters / B, "max_iters": int(out["iters"].max().item()),
           "mean_support": float(out["alen"].float().mean().item()),
           "converged_frac": float(ok.float().mean().item()),
           "status_counts": torch.bincount(out["status"].long(), minlength=5).tolist(),
           "max_marginal_err_all": float(err.max().item()),
           "max_marginal_err_converged": float(err[ok].max().item()) if bool(ok.any()) else None,
           "max_abs_sum_p_minus_1_all": float((out["p"].double().sum(-1) - 1).abs().max().item())}
    if _dist_on():  # whole-job figures: every rank solves its own B samples, slowest rank counts
        ms_max = max_over_ranks(ms, world, dev)
        res.update({"n_gpus": world, "B_total": B * world, "fwd_ms_max_over_ranks": ms_max,
                    "samples_per_s": B * world / (ms_max * 1e-3)})
    return res


def sparsemap_seq_bench(native, dev, B, n, sigma, reps=20, seed=77):
    """The binary sequence factor (sequence_smap) on the same solver: node scores
    and edge potentials ~ N(0, sigma^2)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, n, device=dev, generator=g) * sigma
    t = torch.randn(B, n - 1, device=dev, generator=g) * sigma
    init = torch.rand(B, 2 * n, dtype=torch.double,
                      generator=torch.Generator().manual_seed(42)).to(dev)
    out = native.sparsemap_seq_fwd(x, t, init, 300)
    ms = median_ms(lambda: native.sparsemap_seq_fwd(x, t, init, 300), reps, 2)
    dp = torch.randn(B, n + 1, device=dev, generator=g)
    bms = median_ms(lambda: native.sparsemap_seq_bwd(dp, out["aset_bits"], out["alen"], out["inv_a"], n), reps, 2)
    iters = float(out["iters"].double().sum().item())
    return {"factor": "sequence_binary", "B": B, "positions": n, "sigma": sigma, "fwd_ms": ms, "bwd_ms": bms,
            "launches": reps, "samples_per_s": B / (ms * 1e-3), "active_set_iters_per_s": iters / (ms * 1e-3),
            "mean_iters": iters / B, "mean_support": float(out["alen"].float().mean().item()),
            "max_iters": int(out["iters"].max().item()),
            "converged_frac": float((out["status"] == 0).float().mean().item()),
            "status_counts": torch.bincount(out["status"].long(), minlength=5).tolist(),
            "max_abs_sum_p_minus_1_all": float((out["p"].double().sum(-1) - 1).abs().max().item())}


def compaction_bench(native, dev, R, K, sigma):
    """Kernel 2 on the probabilities of a forward pass, every launch through the C ABI on
    preallocated buffers (median of 20): support_count, support_offsets, support_compact, their
    list-fed forms and the weighted loss reduction, each against the bytes it has to move
    (count 4KR read + 4R; compact 4KR + 12N + 8(R+1); list-fed 4 cap R + 12N +
    8(R+1) + one 64-byte sector per gathered probability; reduce 8N read of vals and comps)."""
    import ctypes
    x, _ = gen(R, K, sigma, dev, 4321)
    p, supp, nnz, _, _, lists, vlists = native.sparsemax_fwd(x, want_nnz=True, want_cols=True, want_vals=True)
    L = native.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    cap = lists.shape[1]
    i64, i32 = ctypes.c_int64, ctypes.c_int32
    # what the value lists cost the forward (same launch with and without them)
    mb = L.smlvm_sparsemax_support_mask_bytes(i64(R), i32(K))
    mask = torch.zeros(max(mb // 4, 1), dtype=torch.int32, device=dev)
    fw = lambda vl: (lambda: L.smlvm_sparsemax_fwd_lists(P(x), P(p), P(supp), None, None, None, P(lists), P(vl),
                                                          i32(cap), P(mask) if mb else None, i64(R), i32(K), st))
    t_fwd_cols, t_fwd_lists = median_ms(fw(None)), median_ms(fw(vlists))
    del x
    off = torch.empty(R + 1, dtype=torch.int64, device=dev)
    ws_b = L.smlvm_support_offsets_workspace_bytes(i64(R))
    ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
    nnz2 = torch.empty_like(nnz)

    def ok(rc):
        assert rc == 0, rc

    t_count = median_ms(lambda: ok(L.smlvm_support_count(P(p), P(nnz2), i64(R), i32(K), st)))
    t_count_l = median_ms(lambda: ok(L.smlvm_support_count_cols(P(p), P(lists), i32(cap), P(nnz2), i64(R), i32(K), st)))
    t_off = median_ms(lambda: ok(L.smlvm_support_offsets(P(nnz), P(off), i64(R), P(ws), ctypes.c_size_t(ws_b), st)))
    N = int(off[R].item())
    rows = torch.empty(N, dtype=torch.int32, device=dev)
    cols = torch.empty(N, dtype=torch.int32, device=dev)
    vals = torch.empty(N, dtype=torch.float32, device=dev)
    t_comp = median_ms(lambda: ok(L.smlvm_support_compact(P(p), P(off), P(rows), P(cols), P(vals), i64(R), i32(K), st)))
    t_comp_l = median_ms(lambda: ok(L.smlvm_support_compact_cols(P(p), P(lists), i32(cap), P(off), P(rows), P(cols), P(vals), i64(R), i32(K), st)))
    # the single pass: count + look-back scan + compaction in one launch, entries sized by R * cap
    ncap = R * cap
    f_rows = torch.empty(ncap, dtype=torch.int32, device=dev)
    f_cols = torch.empty(ncap, dtype=torch.int32, device=dev)
    f_vals = torch.empty(ncap, dtype=torch.float32, device=dev)
    fb = L.smlvm_support_compact_fused_workspace_bytes(i64(R))
    fws = torch.empty(fb // 8, dtype=torch.int64, device=dev)
    fused = lambda vl: (lambda: ok(L.smlvm_support_compact_fused(P(p), P(lists), P(vl), i32(cap), P(off), P(f_rows),
                                                                  P(f_cols), P(f_vals), i64(ncap), i64(R), i32(K),
                                                                  P(fws), ctypes.c_size_t(fb), st)))
    t_fused, t_fused_v = median_ms(fused(None)), median_ms(fused(vlists))
    assert int(fws[2].item()) == N and int(fws[1].item()) == 0
    del f_rows, f_cols, f_vals
    comps = torch.rand(N, device=dev) * 500
    total = torch.empty(1, dtype=torch.float32, device=dev)
    wr_b = L.smlvm_weighted_loss_reduce_workspace_bytes(i64(N))
    wr = torch.empty(wr_b, dtype=torch.uint8, device=dev)
    t_red = median_ms(lambda: ok(L.smlvm_weighted_loss_reduce_fwd(
        P(vals), P(comps), None, None, P(total), ctypes.c_float(1.0 / R), i64(0), i64(N), P(wr),
        ctypes.c_size_t(wr_b), st)))
    # what the Python modules pay end to end: count + offsets + ONE host sync (sizes the lists) + compact
    t_e2e = median_ms(lambda: native.support_compact(p, support_cols=lists, support_vals=vlists), 10, 2)
    t_e2e_chain = median_ms(lambda: native.support_compact(p, support_cols=lists, fused=False), 10, 2)
    listed = float((lists[:, 0] >= -1).float().mean().item())

    def entry(ms, nbytes):
        g = nbytes / (ms * 1e-3) / 1e9
        return {"ms": ms, "bytes": nbytes, "GBps": g, "frac": g / HBM_PEAK_GBS}

    return {"R": R, "K": K, "sigma": sigma, "N": N, "rows_with_list": listed, "launches": 20,
            "support_count": entry(t_count, 4.0 * K * R + 4.0 * R),
            "support_offsets": entry(t_off, 4.0 * R + 8.0 * (R + 1)),
            "support_compact": entry(t_comp, 4.0 * K * R + 12.0 * N + 8.0 * (R + 1)),
            "support_count_cols": entry(t_count_l, listed * 4.0 * cap * R + (1 - listed) * 4.0 * K * R + 4.0 * R),
            "support_compact_cols": entry(t_comp_l, listed * (4.0 * cap * R + 64.0 * N) + (1 - listed) * 4.0 * K * R
                                          + 12.0 * N + 8.0 * (R + 1)),
            "fused_compact_lists": entry(t_fused, listed * (4.0 * cap * R + 64.0 * N) + (1 - listed) * 8.0 * K * R
                                         + 12.0 * N + 8.0 * (R + 1)),
            "fused_compact_lists_vals": entry(t_fused_v, listed * 8.0 * cap * R + (1 - listed) * 8.0 * K * R
                                              + 12.0 * N + 8.0 * (R + 1)),
            "weighted_loss_reduce_fwd": entry(t_red, 8.0 * N),
            "python_support_compact_from_lists_ms": t_e2e,
            "python_support_compact_from_lists_chain_ms": t_e2e_chain,
            "sparsemax_fwd_cols_ms": t_fwd_cols, "sparsemax_fwd_lists_ms": t_fwd_lists,
            "note": "frac = bytes the launch has to move / time / 8 TB/s; the list-fed forms read the "
                    "forward's support lists instead of p (64-byte sector per gathered probability)"}


def train_steps_bench(dev, world, rank):
    """One optimisation step (forward, backward, gradient all-reduce when N > 1, Adam) of each
    BASELINE.json model config on synthetic data of the experiment's shape (bench_models.py)."""
    import bench_models as bm

    def run(model, batch, B, steps=5, warm=2, lr=1e-3):
        opt = torch.optim.Adam(model.parameters(), lr=lr)
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
                "loss": float(out["loss"].detach())}

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
        m, b = bm.make_signal_game(dev, 64, compact=True, hoist=True)
        res["cfg2_signal_game_marg_sparsemax_K256_b64_compact_hoisted"] = run(m, b, 64)
        # the sender above collapses to one message per image within the warm-up steps (support 1:
        # nothing to marginalise, nothing to hoist); the same three steps with the learning rate at
        # zero keep the random-init sender's support, which is what the decoder modes differ on
        for tag, kw in (("dense", {}), ("compact", {"compact": True}),
                        ("compact_hoisted", {"compact": True, "hoist": True})):
            m, b = bm.make_signal_game(dev, 64, **kw)
            res[f"cfg2_signal_game_K256_b64_init_support_{tag}"] = run(m, b, 64, lr=0.0)
        # ... and at a batch where the decoder's work shows above the launch latencies of the step
        for tag, kw in (("compact", {"compact": True}), ("compact_hoisted", {"compact": True, "hoist": True})):
            m, b = bm.make_signal_game(dev, 512, **kw)
            res[f"cfg2_signal_game_K256_b512_init_support_{tag}"] = run(m, b, 512, lr=0.0)
        m, b = bm.make_bitvae(dev, 64, 32, "sparsemap")
        res["cfg4_bitvae_sparsemap_32bits_b64"] = run(m, b, 64)
        m, b = bm.make_bitvae(dev, 64, 32, "sparsemap", fused_decoder=False)
        res["cfg4_bitvae_sparsemap_32bits_b64_dense_logits"] = run(m, b, 64)
        m, b = bm.make_bitvae(dev, 64, 32, "topksparse")
        res["cfg4_bitvae_topksparse_32bits_b64"] = run(m, b, 64)
        del m, b
    # cfg5: global batch 8192 over 8 GPUs = 1024 per rank, one flat-bucket RCCL all-reduce per step
    m, b = bm.make_bitvae(dev, 1024, 128, "sparsemap", seed=0)
    if _dist_on():
        m.encoder.shard = "auto"   # init scores from the global-batch stream (every rank seeds alike)
    res["cfg5_bitvae_sparsemap_128bits_b1024_per_gpu"] = run(m, b, 1024, steps=5, warm=3)
    res["cfg5_bitvae_sparsemap_128bits_b1024_per_gpu"]["grad_allreduce"] = (
        f"1 bucket of {sum(p.numel() for p in m.parameters()) * 4 / 1e6:.1f} MB over {world} rank(s)")
    return res


def make_digest(out, a):
    """A few hundred characters: per sweep entry [rows/s, fwd_frac, bwd_frac, fwd_bwd_frac] (fractions
    of the 8 TB/s peak on the least bytes the launches have to move), the headline's rate on
    SURVEY.md's 20 K streaming model, the compaction chain and the solver's convergence."""
    r3 = lambda v: None if v is None else float(f"{v:.3g}")
    k = out["kernels"]
    d = {"sweep_frac": {f"K{a.K}" if a.sigma == 1.0 else f"K{a.K}_s{a.sigma}":
                        [r3(a.rows / ((k["sparsemax_fwd"]["ms"] + k["sparsemax_bwd"]["ms"]) * 1e-3)),
                         r3(k["sparsemax_fwd"]["frac"]), r3(k["sparsemax_bwd"]["frac"]), r3(out["fwd_bwd_frac"])]},
         "rate_on_20K_model_GBps": r3(20.0 * a.K * a.rows / ((k["sparsemax_fwd"]["ms"] + k["sparsemax_bwd"]["ms"]) * 1e-3) / 1e9),
         "rate_on_20K_model_note": "may exceed the 8000 GB/s peak: the list backward reads O(support) "
                                   "bytes instead of the dy and p rows"}
    for tag, e in out.get("sweep", {}).items():
        d["sweep_frac"][tag.replace("sigma", "s").replace("adversarial", "adv")] = [
            r3(e["rows_per_s"]), r3(e["fwd_frac"]), r3(e["bwd_frac"]), r3(e["fwd_bwd_frac"])]
    if out.get("compaction"):
        d["compaction_ms"] = {f'K{c["K"]}_s{c["sigma"]}': {kk: r3(c[kk]["ms"]) if isinstance(c[kk], dict) else r3(c[kk])
                                                          for kk in c if kk.startswith("support_") or kk.startswith("fused") or kk.startswith("python_")
                              or kk.startswith("sparsemax_fwd_")}
                              for c in out["compaction"]}
    sm = out.get("sparsemap") or []
    if sm:
        d["sparsemap_converged_frac_min"] = r3(min(e["converged_frac"] for e in sm))
        d["sparsemap_converged_frac_min_bernoulli"] = r3(min([e["converged_frac"] for e in sm if e["budget"] == 0] or [None]))
        d["sparsemap_max_marginal_err_all"] = r3(max(e["max_marginal_err_all"] for e in sm))
        d["sparsemap_max_marginal_err_converged"] = r3(max(e["max_marginal_err_converged"] or 0.0 for e in sm))
        d["sparsemap_ms"] = {f'{e["B"]}x{e["bits"]}_b{e["budget"]}_s{e["sigma"]:g}': r3(e["fwd_ms"]) for e in sm}
    sq = out.get("sparsemap_sequence") or []
    if sq:
        d["sequence_converged_frac"] = {f'{e["B"]}x{e["positions"]}': r3(e["converged_frac"]) for e in sq}
    ts = out.get("train_steps") or {}
    if ts:
        d["train_step_ms"] = {kk: r3(v["ms_per_step"]) for kk, v in ts.items()}
    return d


def tail_digest(d):
    """The digest cut down to what fits the driver's tail of the line: [fwd, bwd, fwd+bwd] fractions
    per sweep entry, list-fed compaction end to end, solver convergence, cfg5 step."""
    t = {"frac_fwd_bwd_both": {k: v[1:] for k, v in d["sweep_frac"].items()},
         "rate_on_20K_model_GBps": d["rate_on_20K_model_GBps"]}
    if "compaction_ms" in d:
        t["compact_from_lists_ms_kernel_e2e"] = {k: [v.get("fused_compact_lists_vals"), v.get("python_support_compact_from_lists_ms")]
                                                 for k, v in d["compaction_ms"].items()}
    for k in ("sparsemap_converged_frac_min", "sparsemap_converged_frac_min_bernoulli",
              "sparsemap_max_marginal_err_all", "sparsemap_max_marginal_err_converged", "sequence_converged_frac"):
        if k in d:
            t[k.replace("sparsemap_", "sm_")] = d[k]
    if "sparsemap_ms" in d:
        t["sm_ms"] = {k: v for k, v in d["sparsemap_ms"].items() if k.endswith("_s1") and "x128" in k}
    if "train_step_ms" in d:
        t["step_ms"] = {k.split("_")[0] + ("_compact" if "compact" in k else ""): v for k, v in d["train_step_ms"].items()
                        if "init_support" not in k and "dense_logits" not in k and "hoisted" not in k and "topk" not in k}
    return t


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))     # parent of the N ranks: no GPU call in here
    world, rank, local = dist_setup(a.gpus)
    if world != a.gpus and not (a.gpus == 1 and world == 1):
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local if _dist_on() else 0)
    from lvmhelpers import _native as native
    native.lib()

    x, dy = gen(a.rows, a.K, a.sigma, dev, 1234 + rank)
    tm = timed_sparsemax(native, x, dy, a.steps, a.warmup, world, dev)
    wall = max_over_ranks(tm["wall"], world, dev)
    f_ms, b_ms = tm["fwd_ms"], tm["bwd_ms"]
    del x, dy
    torch.cuda.empty_cache()

    # proof that N ranks took part and that their shards differ (rank r generates seed 1234 + r,
    # so rank 0's checksums are the same at every N and equal the 1-GPU run's)
    ranks_seen, checks, rccl = 1, [[tm["checksum_p"], tm["checksum_dx"]]], None
    if _dist_on():
        import torch.distributed as dist
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        mine = torch.tensor([tm["checksum_p"], tm["checksum_dx"]], dtype=torch.int64, device=dev)
        allc = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        checks = [[int(v) for v in t.tolist()] for t in allc]
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = None

    rows_total = a.rows * world * a.steps
    fwd_alg = 8.0 * a.K * a.rows               # read x + write p
    mt = tm["min_traffic"]
    gbps = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9
    kern = {
        "sparsemax_fwd": {"ms": f_ms, "ms_median": tm["fwd_ms_median"], "algorithmic_bytes": fwd_alg,
                          "GBps": gbps(fwd_alg, f_ms), "min_traffic_bytes": mt["fwd"],
                          "frac": gbps(mt["fwd"], f_ms) / HBM_PEAK_GBS},
        "sparsemax_bwd": {"ms": b_ms, "ms_median": tm["bwd_ms_median"], "min_traffic_bytes": mt["bwd"],
                          "GBps": gbps(mt["bwd"], b_ms), "frac": gbps(mt["bwd"], b_ms) / HBM_PEAK_GBS,
                          "streaming_model_bytes": 12.0 * a.K * a.rows,
                          "note": "GBps and frac are on the bytes this launch has to move (dx written, "
                                  "support list + one 64-byte sector of dy per support element read), "
                                  "not on SURVEY.md's 12 K streaming model, which it undercuts by "
                                  "reading neither p nor the dy row"},
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
    dom_bytes = fwd_alg if dom == "sparsemax_fwd" else mt["bwd"]
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
                   "rows_per_gpu": a.rows, "K": a.K, "mean_support": tm["mean_support"],
                   "parallelism": f"rows sharded over {world} GPU(s), no collective"},
        "ranks_seen": ranks_seen,
        "rank_checksums_p_dx": checks,
        "rccl_version": rccl,
        "fwd_bwd_min_traffic_GBps_per_gpu": gbps(mt["fwd"] + mt["bwd"], f_ms + b_ms),
        "fwd_bwd_frac": gbps(mt["fwd"] + mt["bwd"], f_ms + b_ms) / HBM_PEAK_GBS,
        "backward_rows": mt["rows"],
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": gbps(dom_bytes, kern[dom]["ms"]),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": gbps(dom_bytes, kern[dom]["ms"]) / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "note": "dominant kernel; achieved = algorithmic bytes per launch (forward: 8 K per "
                             "row, SURVEY.md 8(d)) / mean launch time by HIP events in the timed region. "
                             "`fwd_bwd_frac` and kernels.*.frac use the least bytes each launch has to "
                             "move (min_traffic in bench.py), so no fraction here can exceed 1."},
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
            t = timed_sparsemax(native, xs, dys, 20, 5, 1, dev)   # 8(d): >= 20 launches, median
            del xs, dys
            torch.cuda.empty_cache()
            fm, bm_ = t["fwd_ms_median"], t["bwd_ms_median"]
            m = t["min_traffic"]
            sweep[tag] = {"rows_per_launch": R, "launches_per_1M_rows": chunks, "launches": 20,
                          "fwd_ms": fm, "bwd_ms": bm_, "rows_per_s": R / ((fm + bm_) * 1e-3),
                          "fwd_frac": gbps(m["fwd"], fm) / HBM_PEAK_GBS,
                          "bwd_frac": gbps(m["bwd"], bm_) / HBM_PEAK_GBS,
                          "fwd_bwd_frac": gbps(m["fwd"] + m["bwd"], fm + bm_) / HBM_PEAK_GBS,
                          "min_traffic_bytes": m["fwd"] + m["bwd"],
                          "rate_on_20K_model_GBps": gbps(20.0 * K * R, fm + bm_),
                          "backward_rows": m["rows"], "mean_support": t["mean_support"]}

        # K in {256, 4096, 65536} x {N(0,1), N(0,0.05^2), adversarial}; the
        # headline line above is (4096, N(0,1)).  Fractions are on the least bytes the two launches
        # have to move (see min_traffic); rate_on_20K_model_GBps is SURVEY.md's streaming model and
        # may exceed the HBM peak because the list backward reads neither p nor dy.
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
            t = timed_sparsemax(native, xs, dys, 200, 20, 1, dev)
            small[f"{R}x{K}"] = {"fwd_us": t["fwd_ms"] * 1e3, "bwd_us": t["bwd_ms"] * 1e3,
                                 "pair_wall_us": t["wall"] / 200 * 1e6}
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
        # sigma in {1, 3} x (B, n) in {(64, 32), (8192, 128)} x budget in {0, n/2}
        # (+ the 1024-sample per-GPU shard of config 5)
        out["sparsemap"] = [sparsemap_bench(native, dev, B, n, bud, sg)
                            for (B, n) in ((64, 32), (1024, 128), (8192, 128))
                            for bud in (0, n // 2) for sg in (1.0, 3.0)]
        out["sparsemap_sequence"] = [sparsemap_seq_bench(native, dev, 64, 32, 1.0),
                                     sparsemap_seq_bench(native, dev, 1024, 128, 1.0)]
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
        out["sparsemap_cpu_baseline"] = sparsemap_cpu_baseline(8192, 128, 0, 1.0)   # ~5 s on one core
        out["sparsemap_cpu_baseline_all_cores"] = sparsemap_cpu_baseline_all_cores(32768, 128, 0, 1.0)
    if rank == 0:
        # The driver's record keeps `config`, `roofline`, `cpu_baseline` whole and otherwise only the
        # last ~2000 characters of this line: the compact digest of the sweep / compaction / solver
        # blocks goes into `roofline` AND to the very end of the line.
        digest = make_digest(out, a)
        out["roofline"]["digest"] = digest
        out["digest_tail"] = tail_digest(digest)   # last key of the line, < 1500 characters
        print(json.dumps(out))
    if _dist_on():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
