"""GPU box: forward / list-backward launch times per adversarial row KIND (all rows of one kind):
python tools/t_adv_kinds.py [K R]..."""
import sys
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
import os
if os.environ.get('SMLVM_T_LIB'):   # A/B runs: another build of the library
    _native.LIB_PATH = os.environ['SMLVM_T_LIB']

dev = torch.device('cuda:0')


def rows(R, K, kind):
    g = torch.Generator(device=dev).manual_seed(99)
    xs = torch.empty(R, K, device=dev).normal_(0.0, 1.0, generator=g)
    if kind == 0:
        xs[:] = 0.25
    elif kind == 1:
        xs[:] = -1.0
        xs[:, 7 % K] = 3.0
    elif kind == 2:
        xs[:] = -1.0
        xs[:, 0] = 0.0
    elif kind == 4:
        xs *= 0.05
    return xs


def ms(f, n=20, w=5):
    for _ in range(w):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


NAMES = {0: "all-equal", 1: "dominant", 2: "tie-at-threshold", 3: "N(0,1)", 4: "N(0,0.05^2)"}
args = [int(a) for a in sys.argv[1:]] or [256, 1 << 20]
for K, R in zip(args[::2], args[1::2]):
    for kind in range(5):
        x = rows(R, K, kind)
        dy = torch.randn_like(x)
        out = _native.sparsemax_fwd(x, want_cols=True, want_mask=True)
        p, supp, cols, mask = out[0], out[1], out[5], out[6]
        tf = ms(lambda: _native.sparsemax_fwd(x, want_cols=True, want_mask=True))
        tb = ms(lambda: _native.sparsemax_bwd_cols(dy, cols, p, supp, mask))
        print(f"K={K} R={R} {NAMES[kind]:18s} fwd {tf:.4f} ms  bwd {tb:.4f} ms", flush=True)
        del x, dy, out, p, supp, cols, mask
