"""GPU box: forward / list backward at 32768 x 65536 and 131072 x 16384 (segment kernel), N(0,1) and N(0,0.05^2)."""
import os, sys, torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
if len(sys.argv) > 1:
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device('cuda:0')
for R, K in ((32768, 65536), (131072, 16384), (65536, 8192 + 4)):
    for sigma in (1.0, 0.05):
        g = torch.Generator(device=dev).manual_seed(99)
        x = torch.empty(R, K, device=dev).normal_(0.0, sigma, generator=g)
        dy = torch.randn(R, K, device=dev, generator=g)
        out = _native.sparsemax_fwd(x, want_cols=True, want_mask=True)
        p, supp, cols, mask = out[0], out[1], out[5], out[6]
        f = lambda: _native.sparsemax_bwd_cols(dy, cols, p, supp, mask)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"R {R} K {K} sigma {sigma}: bwd {e0.elapsed_time(e1) / 10:.3f} ms  mean support {float(supp.float().mean()):.1f}", flush=True)
        del x, dy, out, p, supp, cols, mask
