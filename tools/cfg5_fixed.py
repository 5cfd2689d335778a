"""Config-5 step at a FIXED model state (lr = 0): python tools/cfg5_fixed.py [pkgdir] -- compares
package versions without the step time drifting with training."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sparse-marginalization-lvm_amd"))
import torch
import bench_models as bm
import lvmhelpers
print(lvmhelpers.__file__)
dev = torch.device("cuda", 0)
model, batch = bm.make_bitvae(dev, 1024, 128, "sparsemap", seed=0)
opt = torch.optim.Adam(model.parameters(), lr=0.0)
for _ in range(3):
    out = bm.train_step(model, opt, batch)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        out = bm.train_step(model, opt, batch)
    torch.cuda.synchronize()
    print("ms per step %.3f" % ((time.perf_counter() - t0) / 10 * 1e3), "N", len(out["log"]["idxs"]))
