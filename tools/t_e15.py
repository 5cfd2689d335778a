"""GPU box: entmax15 forward/backward launch times: python tools/t_e15.py [K R sigma]..."""
import sys
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
dev = torch.device('cuda:0')


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


args = sys.argv[1:] or ["4096", str(1 << 20), "1.0"]
for K, R, sg in zip(args[::3], args[1::3], args[2::3]):
    K, R, sg = int(K), int(R), float(sg)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(R, K, device=dev, generator=g) * sg
    dy = torch.randn(R, K, device=dev, generator=g)
    p, nnz, _ = _native.entmax15_fwd(x, want_nnz=True)
    tf = ms(lambda: _native.entmax15_fwd(x))
    tb = ms(lambda: _native.entmax15_bwd(dy, p))
    ts = ms(lambda: _native.sparsemax_fwd(x))
    print(f"K={K} R={R} sigma={sg}: entmax15 fwd {tf:.3f} ms bwd {tb:.3f} ms (support {nnz.float().mean().item():.1f}); sparsemax fwd {ts:.3f} ms", flush=True)
    del x, dy, p
