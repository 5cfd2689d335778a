"""GPU box: the p-based backward (smlvm_sparsemax_bwd: dy, p, supp -> dx) at 1M x 4096:
python tools/t_bwd_p.py [lib.so]"""
import os, sys, torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
if len(sys.argv) > 1:
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device('cuda:0')
R, K = 1 << 20, 4096
for sigma in (1.0, 0.2, 0.05):
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.empty(R, K, device=dev).normal_(0.0, sigma, generator=g)
    dy = torch.randn(R, K, device=dev, generator=g)
    out = _native.sparsemax_fwd(x)
    p, supp = out[0], out[1]
    f = lambda: _native.sparsemax_bwd(dy, p, supp)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{(sys.argv[1:] or ['tree'])[0]:26s} sigma {sigma}: p-based bwd {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
    del x, dy, out, p, supp
