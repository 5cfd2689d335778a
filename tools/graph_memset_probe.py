"""GPU box: memset nodes in HIP graph replays (PyTorch 2.10.0+rocm7.0, bundled HIP 7.0.51831).
A column sum of a [640, 3920] matrix makes at::native::reduce_kernel's launcher zero its semaphores
with cudaMemsetAsync; captured in a graph (plain torch.cuda.graph, none of this repository's
kernels) the sum is right on the first replay and wrong on the later ones.  With the captured graph
passed through smlvm_graph_memsets_to_kernels (lvmhelpers.graphs.capture) every replay is right.
The second part is the smallest case: a graph of [hipMemsetAsync(buf, 0)] [out = buf]; replay 0
reads 0, later replays read an arbitrary non-zero word -- the node runs, with a value that is no
longer the captured one.
    python tools/graph_memset_probe.py"""
import sys
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
from lvmhelpers import graphs
dev = torch.device('cuda:0')
torch.manual_seed(0)
status = 0
for rows, cols in ((640, 3920), (2112, 3920), (64, 3920), (16384, 256)):
    for repaired in (False, True):
        x = torch.randn(rows, cols, device=dev)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                (x * 2).sum(0)
        torch.cuda.current_stream().wait_stream(s)
        box = {}

        def body():
            a = x * 2
            b = torch.relu(a) + 1        # temporaries that die before the reduction takes its buffers
            del b
            box['y'] = a.sum(0)
        n_fixed = None
        if repaired:
            g, n_fixed = graphs.capture(body)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
        bad = 0
        for i in range(20):
            x.copy_(torch.randn(rows, cols, device=dev))
            g.replay()
            torch.cuda.synchronize()
            err = float((box['y'] - (x * 2).sum(0)).abs().max())
            bad += 0 if err <= 1e-3 else 1
        print(f"[{rows} x {cols}] sum(0), {'repaired graph (' + str(n_fixed) + ' memset nodes replaced)' if repaired else 'plain torch.cuda.graph'}: "
              f"{bad} wrong replays of 20", flush=True)
        if repaired and bad:
            status = 1

import ctypes
hip = ctypes.CDLL('libamdhip64.so')          # (the runtime already loaded by torch)
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
buf = torch.empty(64, dtype=torch.int32, device=dev)
out = torch.empty_like(buf)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    buf.fill_(1)
    out.copy_(buf)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    assert hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * 4, torch.cuda.current_stream().cuda_stream) == 0
    out.copy_(buf)
for i in range(3):
    buf.fill_(7)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print(f"graph [memset buf = 0][out = buf], buf = 7 before the replay: replay {i} reads {int(out[0])}", flush=True)
sys.exit(status)
