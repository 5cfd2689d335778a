"""Round 3: the `compaction` entries of bench.py alone (quick GPU check)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-marginalization-lvm_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from lvmhelpers import _native as native
dev = torch.device("cuda", 0)
native.lib()
for (R, K, s) in ((1 << 20, 4096, 1.0), (1 << 20, 256, 1.0), (1 << 20, 256, 0.05)):
    r = bench.compaction_bench(native, dev, R, K, s)
    print(json.dumps({k: (round(v["ms"], 4) if isinstance(v, dict) else v) for k, v in r.items() if k != "note"}), flush=True)
