"""GPU box: fwd_cols / bwd_cols launch times with the mask buffer for short rows."""
import sys
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
dev = torch.device('cuda:0')
def ms(f, n=20, w=5):
    for _ in range(w): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for K, sg in ((256, 1.0), (256, 0.05), (512, 0.05), (1024, 0.05), (128, 0.05)):
    R = 1 << 20
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.empty(R, K, device=dev).normal_(0, sg, generator=g)
    dy = torch.empty(R, K, device=dev).normal_(0, 1, generator=g)
    out = {}
    for wm in (False, True):
        r = _native.sparsemax_fwd(x, want_cols=True, want_mask=wm)
        p, supp, cols = r[0], r[1], r[5]
        mask = r[6] if wm else None
        out[wm] = (ms(lambda: _native.sparsemax_fwd(x, want_cols=True, want_mask=wm)),
                   ms(lambda: _native.sparsemax_bwd_cols(dy, cols, p, supp, mask)))
    print(K, sg, "no mask fwd/bwd %.4f %.4f   mask fwd/bwd %.4f %.4f" % (out[False] + out[True]))
