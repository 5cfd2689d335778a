"""GPU box: host-side cost per call of the ctypes wrappers next to a torch op (no synchronisation\ninside the timed loops):  python tools/t_host_overhead.py"""
import sys, time
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native, functional as F_
dev = torch.device('cuda:0')
x = torch.randn(64, 10, device=dev)
p = _native.sparsemax_fwd(x)[0]
def t(f, n=3000):
    for _ in range(200): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
print("torch.softmax            %.1f us" % t(lambda: torch.softmax(x, -1)))
print("torch.empty_like         %.1f us" % t(lambda: torch.empty_like(x)))
print("_native.sparsemax_fwd    %.1f us" % t(lambda: _native.sparsemax_fwd(x)))
print("F_.sparsemax (autograd)  %.1f us" % t(lambda: F_.sparsemax(x, -1)))
print("_native.support_count    %.1f us" % t(lambda: _native.support_count(p)))
print("with torch.cuda.device   %.1f us" % t(lambda: torch.cuda.device(dev).__enter__()))
print("current_stream().cuda_stream %.1f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("x.data_ptr()             %.1f us" % t(lambda: x.data_ptr()))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(2000): _native.sparsemax_fwd(x)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
