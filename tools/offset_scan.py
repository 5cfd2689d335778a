"""GPU box: headline forward / list backward times as a function of the RELATIVE placement of their
tensors inside one arena (one process, one physical placement): p at x + 16 GiB + d, dx at dy + 16 GiB + d.
python tools/offset_scan.py"""
import ctypes
import sys
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native as native

dev = torch.device('cuda:0')
R, K = 1 << 20, 4096
n = R * K
pad = 1 << 28                      # floats of slack after each tensor (1 GiB)
arena = torch.empty(4 * (n + pad), dtype=torch.float32, device=dev)
base = [i * (n + pad) for i in range(4)]
x = arena[base[0]:base[0] + n].view(R, K)
dy = arena[base[2]:base[2] + n].view(R, K)
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


def timed(f, reps=4):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[reps // 2]


print(f"arena {arena.data_ptr():#x}", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "abs":
    # ABSOLUTE position: all four tensors move together by a (x, p, dy, dx keep their distances)
    del x, dy
    big = torch.empty(4 * n + 48 * (1 << 28) + (1 << 20), dtype=torch.float32, device=dev)   # + 48 GiB of room
    del arena
    print(f"arena2 {big.data_ptr():#x}", flush=True)
    for a_mb in (0, 10240, 0, 4096, 20480, 8192, 32768, 0, 49152, 16384, 2048, 40960):
        a = a_mb << 18
        xs = big[a:a + n].view(R, K)
        ps = big[a + n:a + 2 * n].view(R, K)
        dys = big[a + 2 * n:a + 3 * n].view(R, K)
        dxs = big[a + 3 * n:a + 4 * n].view(R, K)
        g = torch.Generator(device=dev).manual_seed(1234)
        xs.normal_(0.0, 1.0, generator=g)
        dys.normal_(0.0, 1.0, generator=g)
        fwd = lambda: L.smlvm_sparsemax_fwd_cols(P(xs), P(ps), P(supp), None, None, None, P(cols),
                                                 ctypes.c_int32(cap), mp, ctypes.c_int64(R), ctypes.c_int32(K), st)
        bwd = lambda: L.smlvm_sparsemax_bwd_cols(P(dys), P(cols), ctypes.c_int32(cap), mp, P(ps), P(supp), P(dxs),
                                                 ctypes.c_int64(R), ctypes.c_int32(K), st)
        print(f"a = {a_mb:6d} MiB: x {xs.data_ptr():#x}  fwd {timed(fwd):.3f} ms  bwd {timed(bwd):.3f} ms", flush=True)
    sys.exit(0)
for d_kb in (0, 4, 16, 64, 128, 256, 512, 1024, 1536, 2048, 3072, 4096, 8192, 16384, 65536, 262144, 524288):
    d = d_kb * 256   # floats
    p = arena[base[1] + d:base[1] + d + n].view(R, K)
    dx = arena[base[3] + d:base[3] + d + n].view(R, K)
    fwd = lambda: L.smlvm_sparsemax_fwd_cols(P(x), P(p), P(supp), None, None, None, P(cols), ctypes.c_int32(cap),
                                             mp, ctypes.c_int64(R), ctypes.c_int32(K), st)
    bwd = lambda: L.smlvm_sparsemax_bwd_cols(P(dy), P(cols), ctypes.c_int32(cap), mp, P(p), P(supp), P(dx),
                                             ctypes.c_int64(R), ctypes.c_int32(K), st)
    tf = timed(fwd)
    tb = timed(bwd)
    print(f"d = {d_kb:7d} KiB: p - x = {(p.data_ptr() - x.data_ptr()) / 2**20:10.3f} MiB  fwd {tf:.3f} ms   "
          f"dx - dy = {(dx.data_ptr() - dy.data_ptr()) / 2**20:10.3f} MiB  bwd {tb:.3f} ms", flush=True)
