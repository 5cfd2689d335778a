"""GPU box: SparseMAP forward launch times over the S5 grid and a few seeds:
python tools/t_sm.py [lib.so]"""
import sys
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
if len(sys.argv) > 1:
    _native.LIB_PATH = sys.argv[1]
dev = torch.device('cuda:0')


def med(f, n=10, w=2):
    for _ in range(w):
        f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


for (B, n) in ((64, 32), (1024, 128), (8192, 128)):
    for bud in (0, n // 2):
        for sg in (1.0, 3.0):
            ts, its = [], []
            for seed in (77, 78, 79, 80):
                g = torch.Generator(device=dev).manual_seed(seed)
                x = torch.randn(B, n, device=dev, generator=g) * sg
                init = torch.rand(B, 2 * n, dtype=torch.double, generator=torch.Generator().manual_seed(42 + seed)).to(dev)
                out = _native.sparsemap_fwd(x, init, bud, 300)
                its.append(int(out["iters"].sum().item()))
                ts.append(med(lambda: _native.sparsemap_fwd(x, init, bud, 300)))
            print(f"B={B} n={n} budget={bud} sigma={sg}: " + " ".join(f"{t:.3f}" for t in ts) + f"  mean {sum(ts)/len(ts):.3f} ms  iters {its}", flush=True)
