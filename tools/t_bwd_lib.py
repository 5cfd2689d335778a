"""GPU box: list / pattern backward of the headline shape for the tree's library and others:
python tools/t_bwd_lib.py [lib.so ...]   (N(0,1), N(0,0.2^2), N(0,0.05^2) at 1M x 4096; T_R / T_K in the
environment for another shape)"""
import os, subprocess, sys
if sys.argv[1:2] != ["--one"]:
    for lib in ["tree"] + sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one", lib], check=False)
    sys.exit(0)
sys.argv[1:] = sys.argv[2:]
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
if sys.argv[1] != "tree":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device('cuda:0')
R, K = int(os.environ.get('T_R', 1 << 20)), int(os.environ.get('T_K', 4096))
for sigma in (1.0, 0.2, 0.05):
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.empty(R, K, device=dev).normal_(0.0, sigma, generator=g)
    dy = torch.randn(R, K, device=dev, generator=g)
    out = _native.sparsemax_fwd(x, want_cols=True, want_mask=True)
    p, supp, cols, mask = out[0], out[1], out[5], out[6]
    dx = torch.empty_like(x)
    f = lambda: _native.sparsemax_bwd_cols(dy, cols, p, supp, mask)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{sys.argv[1]:28s} sigma {sigma}: bwd {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
    del x, dy, out, p, supp, cols, mask, dx
