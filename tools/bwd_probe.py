"""GPU box: does the list backward's time depend on what its output buffer held, on idle time, or on
which buffer it writes?  One process, fixed inputs.  python tools/bwd_probe.py"""
import ctypes
import sys
import time
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native as native

dev = torch.device('cuda:0')
R, K = 1 << 20, 4096
x = torch.empty(R, K, device=dev)
dy = torch.empty(R, K, device=dev)
p = torch.empty(R, K, device=dev)
dx = torch.empty(R, K, device=dev)
dx2 = torch.empty(R, K, device=dev)
g = torch.Generator(device=dev).manual_seed(1234)
x.normal_(0.0, 1.0, generator=g)
dy.normal_(0.0, 1.0, generator=g)
supp = torch.empty(R, dtype=torch.int32, device=dev)
cap = native.support_cols_cap(K)
cols = torch.empty((R, cap), dtype=torch.int32, device=dev)
L = native.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
mbytes = L.smlvm_sparsemax_support_mask_bytes(ctypes.c_int64(R), ctypes.c_int32(K))
mask = torch.zeros(mbytes // 4, dtype=torch.int32, device=dev)
mp = ctypes.c_void_p(mask.data_ptr())
P = lambda t: ctypes.c_void_p(t.data_ptr())
assert L.smlvm_sparsemax_fwd_cols(P(x), P(p), P(supp), None, None, None, P(cols), ctypes.c_int32(cap), mp,
                                  ctypes.c_int64(R), ctypes.c_int32(K), st) == 0


def bwd(out):
    assert L.smlvm_sparsemax_bwd_cols(P(dy), P(cols), ctypes.c_int32(cap), mp, P(p), P(supp), P(out),
                                      ctypes.c_int64(R), ctypes.c_int32(K), st) == 0


def t(out, tag):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); bwd(out); e1.record()
    torch.cuda.synchronize()
    print(f"{tag:34s} {e0.elapsed_time(e1):.3f} ms", flush=True)


for i in range(4):
    t(dx, f"dx, launch {i}")
dx.normal_(0.0, 1.0, generator=g)
t(dx, "dx after random fill")
t(dx, "dx again")
dx.zero_()
t(dx, "dx after zero fill")
t(dx, "dx again")
for i in range(3):
    t(dx2, f"dx2 (fresh buffer), launch {i}")
dx2.normal_(0.0, 1.0, generator=g)
t(dx2, "dx2 after random fill")
t(dx2, "dx2 again")
time.sleep(1.0)
t(dx, "dx after 1 s idle")
t(dx, "dx again")
dy2 = dy.clone()
t(dx, "dx after cloning dy")
p.normal_(0.0, 1.0, generator=g)
t(dx, "dx after random fill of p")
t(dx, "dx again")

# which input decides?  the same lists / sizes / gradient at other places
print("-- lists (cols) elsewhere")
pool = torch.empty(3 << 28, dtype=torch.int32, device=dev)   # 3 GiB
for off_mb in (0, 1, 2, 64, 130, 512, 1024, 2048):
    c2 = pool[off_mb << 18:(off_mb << 18) + cols.numel()].view_as(cols)
    c2.copy_(cols)
    save = cols
    cols = c2
    t(dx, f"cols at pool + {off_mb} MiB -> dx")
    t(dx2, f"cols at pool + {off_mb} MiB -> dx2")
    cols = save
print("-- gradient (dy) elsewhere")
dy3 = torch.empty(R, K, device=dev)
dy3.copy_(dy)
save = dy
dy = dy3
t(dx, "dy copy -> dx"); t(dx, "again"); t(dx2, "dy copy -> dx2"); t(dx2, "again")
dy = save
print(f"x {x.data_ptr():#x} dy {dy.data_ptr():#x} p {p.data_ptr():#x} dx {dx.data_ptr():#x} dx2 {dx2.data_ptr():#x} "
      f"cols {cols.data_ptr():#x} supp {supp.data_ptr():#x} pool {pool.data_ptr():#x} dy3 {dy3.data_ptr():#x}")
