"""GPU box: which of the two speed modes of the headline pair a process lands in,
as a function of how its four 17 GB tensors were allocated and of whether launches run back to back.
python tools/mode_probe.py            -> runs every variant in a process of its own
python tools/mode_probe.py <variant>  -> one variant"""
import os
import subprocess
import sys

VARIANTS = ["plain", "sync", "dummy3g", "outputs_first", "expandable", "one_arena", "sleep"]
if len(sys.argv) < 2 or sys.argv[1] == "sweep":
    if len(sys.argv) >= 2:   # placement sweep: a dummy allocation of N MB first, with both allocator modes
        VARIANTS = [f"{a}dummy{n}m" for n in (0, 2, 16, 34, 128, 1000, 2050) for a in ("", "expandable+")]
    for v in VARIANTS:
        env = dict(os.environ)
        if v.startswith("expandable"):
            env["PYTORCH_HIP_ALLOC_CONF"] = "expandable_segments:True"
        subprocess.run([sys.executable, __file__, v], env=env, check=False)
    sys.exit(0)

import ctypes
import time
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import _native as native
if os.environ.get('SMLVM_LIB'):
    native.LIB_PATH = os.path.abspath(os.environ['SMLVM_LIB'])

v = sys.argv[1]
dev = torch.device('cuda:0')
R, K = 1 << 20, 4096
keep = None
if v == "dummy3g":
    keep = torch.empty(3 << 28, dtype=torch.float32, device=dev)
if v.endswith("m") and "dummy" in v:
    mb = int(v[v.index("dummy") + 5:-1])
    if mb:
        keep = torch.empty(mb << 18, dtype=torch.float32, device=dev)


def rnd(t, sigma, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    t.normal_(0.0, sigma, generator=g)


if v == "one_arena":
    arena = torch.empty(4 * R * K, dtype=torch.float32, device=dev)
    x, dy, p, dx = (arena[i * R * K:(i + 1) * R * K].view(R, K) for i in range(4))
elif v == "outputs_first":
    p = torch.empty(R, K, device=dev)
    dx = torch.empty(R, K, device=dev)
    x = torch.empty(R, K, device=dev)
    dy = torch.empty(R, K, device=dev)
else:
    x = torch.empty(R, K, device=dev)
    dy = torch.empty(R, K, device=dev)
    p = torch.empty(R, K, device=dev)
    dx = torch.empty(R, K, device=dev)
rnd(x, 1.0, 1234)
rnd(dy, 1.0, 99)
supp = torch.empty(R, dtype=torch.int32, device=dev)
cap = native.support_cols_cap(K)
cols = torch.empty((R, cap), dtype=torch.int32, device=dev)
L = native.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
mbytes = L.smlvm_sparsemax_support_mask_bytes(ctypes.c_int64(R), ctypes.c_int32(K))
mask = torch.zeros(mbytes // 4, dtype=torch.int32, device=dev)
mp = ctypes.c_void_p(mask.data_ptr())
P = lambda t: ctypes.c_void_p(t.data_ptr())


def fwd():
    assert L.smlvm_sparsemax_fwd_cols(P(x), P(p), P(supp), None, None, None, P(cols), ctypes.c_int32(cap), mp,
                                      ctypes.c_int64(R), ctypes.c_int32(K), st) == 0


def bwd():
    assert L.smlvm_sparsemax_bwd_cols(P(dy), P(cols), ctypes.c_int32(cap), mp, P(p), P(supp), P(dx),
                                      ctypes.c_int64(R), ctypes.c_int32(K), st) == 0


def gap():
    if v == "sync":
        torch.cuda.synchronize()
    elif v == "sleep":
        torch.cuda.synchronize()
        time.sleep(0.002)


for _ in range(3):
    fwd(); gap(); bwd(); gap()
n = 12
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
for i in range(n):
    ev[i][0].record(); fwd(); ev[i][1].record(); gap(); ev[i][2].record(); bwd(); ev[i][3].record(); gap()
torch.cuda.synchronize()
f = sorted(e[0].elapsed_time(e[1]) for e in ev)
b = sorted(e[2].elapsed_time(e[3]) for e in ev)
print(f"{v:24s} fwd {f[n // 2]:.3f} ms  bwd {b[n // 2]:.3f} ms   (x {x.data_ptr():#x} dy {dy.data_ptr():#x} "
      f"p {p.data_ptr():#x} dx {dx.data_ptr():#x})", flush=True)
