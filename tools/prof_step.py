"""GPU box: host-side profile (cProfile) of a model-shaped training step of bench_models.py:
python tools/prof_step.py cfg4|cfg4_topk|cfg1|cfg2  -> top functions by own time and by cumulative time"""
import cProfile
import pstats
import sys
import time
import torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
import bench_models as bm

dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
if which == 'cfg4':
    m, b = bm.make_bitvae(dev, 64, 32, 'sparsemap')
elif which == 'cfg4_topk':
    m, b = bm.make_bitvae(dev, 64, 32, 'topksparse')
elif which == 'cfg1':
    m, b = bm.make_ssvae(dev, 64, compact=True)
else:
    m, b = bm.make_signal_game(dev, 64, compact=True)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
for _ in range(20):
    bm.train_step(m, opt, b, distributed=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 200
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    bm.train_step(m, opt, b, distributed=False)
torch.cuda.synchronize()
pr.disable()
print(f"{which}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step (under cProfile)")
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(45)
