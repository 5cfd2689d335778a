"""GPU box: one captured batch-64 step replayed N times (for a rocprofv3 kernel trace of what a
replay is made of):  python tools/t_graph_step.py cfg4|cfg4_topk|cfg1|cfg2 [replays]"""
import sys
import time
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
import bench_models as bm
from lvmhelpers.graphs import GraphedStep
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if which == 'cfg1':
    m, b = bm.make_ssvae(dev, 64, compact=True)
elif which == 'cfg2':
    m, b = bm.make_signal_game(dev, 64, compact=True, hoist=True)
    m.static_cap = 32
else:
    m, b = bm.make_bitvae(dev, 64, 32, 'topksparse' if 'topk' in which else 'sparsemap')
step = GraphedStep(m, torch.optim.Adam(m.parameters(), lr=1e-30, capturable=True), b)
for _ in range(10):
    step(*b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step(*b)
torch.cuda.synchronize()
print(f"{which}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per replayed step, {step.memsets_replaced} memset nodes replaced", flush=True)
