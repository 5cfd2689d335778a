"""One rank's share of BASELINE.json config 5 (1024 samples, 128 bits): a few optimisation steps of
the bit-vector VAE, for `rocprofv3 --kernel-trace --stats -- python3 tools/cfg5_step.py`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-marginalization-lvm_amd"))
import torch
import bench_models as bm

dev = torch.device("cuda", 0)
model, batch = bm.make_bitvae(dev, 1024, 128, "sparsemap", seed=0)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
for _ in range(3):
    bm.train_step(model, opt, batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(n):
    bm.train_step(model, opt, batch)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / n * 1e3)
