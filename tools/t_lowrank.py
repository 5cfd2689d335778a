"""GPU box: launch times and fp64-referenced errors of the decoder-loss kernels (csrc/lowrank_ce.hip)
at the config-5 size:  python tools/t_lowrank.py [N]   (SMLVM_LIB=<path> times another build)"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native

if os.environ.get('SMLVM_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['SMLVM_LIB'])
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 53000
Fe, C, R = 784, 256, 5
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(N, Fe, R, device=dev, generator=g)
w = torch.randn(C, R, device=dev, generator=g)
b = torch.randn(C, device=dev, generator=g)
y = torch.randint(0, C, (N, Fe), device=dev, generator=g).to(torch.uint8)
gl = torch.randn(N, device=dev, generator=g)


def ms(f, n=10, wu=3):
    for _ in range(wu):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


loss, lse, nll = _native.lowrank_ce_fwd(x, w, b, y)
print(f"N={N}: fwd {ms(lambda: _native.lowrank_ce_fwd(x, w, b, y)):.3f} ms", flush=True)
print(f"  fwd + dx direction {ms(lambda: _native.lowrank_ce_fwd(x, w, b, y, want_dxdir=True)):.3f} ms", flush=True)
_dd = _native.lowrank_ce_fwd(x, w, b, y, want_dxdir=True)[3]
print(f"  dx = gloss * dxdir (torch) {ms(lambda: _dd * gl.view(-1, 1, 1)):.3f} ms", flush=True)
del _dd
print(f"  bwd dx only {ms(lambda: _native.lowrank_ce_bwd(x, w, b, y, lse, gl, True, False)):.3f} ms", flush=True)
print(f"  bwd dw only {ms(lambda: _native.lowrank_ce_bwd(x, w, b, y, lse, gl, False, True)):.3f} ms", flush=True)
print(f"  bwd both    {ms(lambda: _native.lowrank_ce_bwd(x, w, b, y, lse, gl, True, True)):.3f} ms", flush=True)

# errors on the first rows against fp64 torch
n = 64
x64 = x[:n].double().requires_grad_(True)
w64 = w.double().requires_grad_(True)
b64 = b.double().requires_grad_(True)
lv = F.cross_entropy(F.linear(x64, w64, b64).permute(0, 2, 1), y[:n].long(), reduction="none").sum(1)
(lv * gl[:n].double()).sum().backward()
l2, lse2, _ = _native.lowrank_ce_fwd(x[:n].contiguous(), w, b, y[:n].contiguous())
dx, dw, db = _native.lowrank_ce_bwd(x[:n].contiguous(), w, b, y[:n].contiguous(), lse2, gl[:n].contiguous())
print(f"  loss rel err {float(((l2.double() - lv) / lv).abs().max()):.2e}"
      f"  dx {float((dx.double() - x64.grad).abs().max() / x64.grad.abs().max()):.2e}"
      f"  dw {float((dw.double() - w64.grad).abs().max() / w64.grad.abs().max()):.2e}"
      f"  db {float((db.double() - b64.grad).abs().max() / b64.grad.abs().max()):.2e}")
