"""GPU box: forward / list backward ALTERNATING (as bench.py times them), medians of 20, per library:
T_K=256 python tools/t_pair.py [lib.so ...]"""
import os, subprocess, sys
if sys.argv[1:2] != ["--one"]:
    for lib in ["tree"] + sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one", lib], check=False)
    sys.exit(0)
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
import bench
from lvmhelpers import _native
if sys.argv[2] != "tree":
    _native.LIB_PATH = os.path.abspath(sys.argv[2])
dev = torch.device('cuda:0')
R, K = int(os.environ.get('T_R', 1 << 20)), int(os.environ.get('T_K', 256))
for sigma in (1.0, 0.05):
    x, dy = bench.gen(R, K, sigma, dev, 99)
    t = bench.timed_sparsemax(_native, x, dy, 20, 5, 1, dev)
    print(f"{sys.argv[2]:26s} K {K} sigma {sigma}: fwd {t['fwd_ms_median']:.3f}  bwd {t['bwd_ms_median']:.3f} ms", flush=True)
    del x, dy
