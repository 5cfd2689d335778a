"""GPU box: entmax15 plain pair against the list pair (forward lists, list backward):
python tools/t_e15_lists.py [K R sigma]..."""
import sys, torch
sys.path[:0] = [".", "sparse-marginalization-lvm_amd"]
from lvmhelpers import _native
dev = torch.device("cuda:0")


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


args = sys.argv[1:] or ["4096", str(1 << 20), "1.0", "1024", str(1 << 20), "1.0", "4096", str(1 << 20), "0.3", "2048", str(1 << 20), "1.0"]
for K, R, sg in zip(args[::3], args[1::3], args[2::3]):
    K, R, sg = int(K), int(R), float(sg)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(R, K, device=dev, generator=g) * sg
    dy = torch.randn(R, K, device=dev, generator=g)
    p, cols, vals = _native.entmax15_fwd_lists(x)
    listed = float((cols[:, 0] != -2).float().mean())
    print(f"K={K} s={sg}: fwd {ms(lambda: _native.entmax15_fwd(x)):.3f} fwd_lists {ms(lambda: _native.entmax15_fwd_lists(x)):.3f} "
          f"bwd {ms(lambda: _native.entmax15_bwd(dy, p)):.3f} bwd_lists {ms(lambda: _native.entmax15_bwd_lists(dy, p, cols, vals)):.3f} ms, "
          f"listed {listed:.3f}", flush=True)
    del x, dy, p, cols, vals
