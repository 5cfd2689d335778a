"""Round 3: the `sparsemap` / `sparsemap_sequence` entries of bench.py alone (quick GPU check)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-marginalization-lvm_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from lvmhelpers import _native as native
dev = torch.device("cuda", 0)
native.lib()
res = [bench.sparsemap_bench(native, dev, B, n, bud, sg) for (B, n) in ((64, 32), (1024, 128), (8192, 128))
       for bud in (0, n // 2) for sg in (1.0, 3.0)]
res += [bench.sparsemap_seq_bench(native, dev, 64, 32, 1.0), bench.sparsemap_seq_bench(native, dev, 1024, 128, 1.0)]
for r in res:
    print(json.dumps(r), flush=True)
