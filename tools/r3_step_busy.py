"""Round 3: one model-shaped config per process, fixed model state (lr = 0), 200 timed steps:
    python tools/r3_step_busy.py cfg1|cfg1c|cfg2|cfg2c|cfg4|cfg4_topk|cfg5
prints wall ms per step; run under `rocprofv3 --kernel-trace --stats` the kernel_stats total divided
by the step count gives kernels-busy ms per step (tools/r3_step_busy.sh does both and prints the pair)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-marginalization-lvm_amd")]
import torch
import bench_models as bm
dev = torch.device("cuda", 0)
which = sys.argv[1]
steps = 50 if which == "cfg5" else 200
mk = {"cfg1": lambda: bm.make_ssvae(dev, 64), "cfg1c": lambda: bm.make_ssvae(dev, 64, compact=True),
      "cfg2": lambda: bm.make_signal_game(dev, 64), "cfg2c": lambda: bm.make_signal_game(dev, 64, compact=True),
      "cfg4": lambda: bm.make_bitvae(dev, 64, 32, "sparsemap"), "cfg4_topk": lambda: bm.make_bitvae(dev, 64, 32, "topksparse"),
      "cfg5": lambda: bm.make_bitvae(dev, 1024, 128, "sparsemap", seed=0)}[which]
m, b = mk()
opt = torch.optim.Adam(m.parameters(), lr=0.0)
for _ in range(10):
    bm.train_step(m, opt, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    bm.train_step(m, opt, b)
torch.cuda.synchronize()
print(f"STEP {which} steps {steps + 10} wall_ms_per_step {(time.perf_counter() - t0) / steps * 1e3:.3f}", flush=True)
