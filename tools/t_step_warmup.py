"""GPU box: per-step wall times of the config-1 step with a training warm-up against a learning-rate-0
warm-up followed by training steps (the first call on a new support size shows as one slow step):
python tools/t_step_warmup.py"""
import sys, time, torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
import bench_models as bm
dev = torch.device('cuda:0')
def run(make, mode, warm=8, steps=6):
    torch.manual_seed(0)
    m, b = make()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    if mode == 'lr0':
        for g in opt.param_groups: g['lr'] = 0.0
    ts = []
    for i in range(warm + steps):
        if i == warm and mode == 'lr0':
            for g in opt.param_groups: g['lr'] = 1e-3
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = bm.train_step(m, opt, b, distributed=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    sup = out['log'].get('support')
    print(mode, ' '.join(f'{t:.1f}' for t in ts), 'support', float(torch.as_tensor(sup, dtype=torch.float32).mean()))
for mode in ('const', 'lr0', 'const', 'lr0'):
    run(lambda: bm.make_ssvae(dev, 64), mode)
