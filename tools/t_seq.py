"""GPU box: sequence-factor forward launch times: python tools/t_seq.py"""
import sys
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native
dev = torch.device('cuda:0')
for B, n in ((64, 32), (1024, 128), (8192, 128)):
    g = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn(B, n, device=dev, generator=g)
    t = torch.randn(B, n - 1, device=dev, generator=g)
    init = torch.rand(B, 2 * n, dtype=torch.double, generator=torch.Generator().manual_seed(42)).to(dev)
    out = _native.sparsemap_seq_fwd(x, t, init, 300)
    for _ in range(2):
        _native.sparsemap_seq_fwd(x, t, init, 300)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for e0, e1 in ev:
        e0.record(); _native.sparsemap_seq_fwd(x, t, init, 300); e1.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[5]
    print(f"B={B} n={n}: {ms:.3f} ms  iters {int(out['iters'].sum())}  converged {(out['status'] == 0).float().mean().item():.3f}", flush=True)
