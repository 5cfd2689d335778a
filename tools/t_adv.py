"""GPU box: forward/backward launch times on bench.py's adversarial rows (a quarter each: all equal,
one dominant logit, an exact tie at the threshold, N(0,1)):  python tools/t_adv.py [K R]..."""
import sys
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
import os
if os.environ.get('SMLVM_T_LIB'):   # A/B runs: another build of the library
    _native.LIB_PATH = os.environ['SMLVM_T_LIB']

dev = torch.device('cuda:0')


def adversarial(R, K):
    g = torch.Generator(device=dev).manual_seed(99)
    xs = torch.empty(R, K, device=dev).normal_(0.0, 1.0, generator=g)
    kind = torch.arange(R, device=dev) % 4
    xs[kind == 0] = 0.25
    xs[kind == 1] = -1.0
    xs[kind == 1, 7 % K] = 3.0
    xs[kind == 2] = -1.0
    xs[kind == 2, 0] = 0.0
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


args = [int(a) for a in sys.argv[1:]] or [256, 1 << 20]
for K, R in zip(args[::2], args[1::2]):
    x = adversarial(R, K)
    print(K, R, "fwd %.4f ms" % ms(lambda: _native.sparsemax_fwd(x, want_cols=True)))
