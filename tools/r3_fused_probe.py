"""Round 3 probe: where the time of smlvm_support_compact_fused goes (offsets only / entries / value lists)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-marginalization-lvm_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from lvmhelpers import _native as native
if os.environ.get("SMLVM_LIB"):   # an experiment build of the library
    native.LIB_PATH = os.path.abspath(os.environ["SMLVM_LIB"])
dev = torch.device("cuda", 0)
L = native.lib()
i64, i32 = ctypes.c_int64, ctypes.c_int32
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (R, K, s) in ((1 << 20, 4096, 1.0), (1 << 20, 256, 1.0)):
    x, _ = bench.gen(R, K, s, dev, 4321)
    p, supp, nnz, _, _, lists, vlists = native.sparsemax_fwd(x, want_nnz=True, want_cols=True, want_vals=True)
    del x
    cap = lists.shape[1]
    ncap = R * cap
    off = torch.empty(R + 1, dtype=torch.int64, device=dev)
    rows = torch.empty(ncap, dtype=torch.int32, device=dev); cols = torch.empty_like(rows)
    vals = torch.empty(ncap, dtype=torch.float32, device=dev)
    fb = L.smlvm_support_compact_fused_workspace_bytes(i64(R))
    ws = torch.empty(fb // 8, dtype=torch.int64, device=dev)
    def run(r, c, v, vl):
        return lambda: L.smlvm_support_compact_fused(P(p), P(lists), P(vl), i32(cap), P(off), P(r), P(c), P(v),
                                                     i64(ncap), i64(R), i32(K), P(ws), ctypes.c_size_t(fb), st)
    print(R, K, "offsets only", round(bench.median_ms(run(None, None, None, None)), 4),
          "rows+cols", round(bench.median_ms(run(rows, cols, None, None)), 4),
          "rows+cols+vals(gather)", round(bench.median_ms(run(rows, cols, vals, None)), 4),
          "rows+cols+vals(lists)", round(bench.median_ms(run(rows, cols, vals, vlists)), 4),
          "vals only (lists)", round(bench.median_ms(run(None, None, vals, vlists)), 4),
          "memset alone", round(bench.median_ms(lambda: ws.zero_()), 4), flush=True)
