import sys, json, torch
sys.path[:0] = ['.', 'sparse-marginalization-lvm_amd']
import bench
from lvmhelpers import _native
dev = torch.device('cuda:0')
for R, K, s in ((1 << 20, 4096, 1.0), (1 << 20, 256, 1.0), (1 << 20, 256, 0.05)):
    r = bench.compaction_bench(_native, dev, R, K, s)
    print(json.dumps(r))
