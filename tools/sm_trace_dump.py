"""Writes one tools/sm_trace.c input file per sample of a synthetic batch that does not end
CONVERGED (round 3 diagnosis; CPU only, uses the oracle):  python tools/sm_trace_dump.py B n sigma budget outdir"""
import os, struct, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle

B, n, sigma, budget, out = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
torch.manual_seed(1234)
x = (torch.randn(B, n) * sigma).numpy().astype(np.float32)
init = torch.rand(B, 2 * n, dtype=torch.double, generator=torch.Generator().manual_seed(42)).numpy()
o = oracle.sparsemap_fwd(x, init, budget=budget, max_iter=300)
bad = np.nonzero(o["status"] != 0)[0]
print("status counts", np.bincount(o["status"], minlength=5), "not converged:", bad)
os.makedirs(out, exist_ok=True)
for b in bad:
    with open(os.path.join(out, f"s{budget}_{b}.bin"), "wb") as f:
        f.write(struct.pack("iii", n, budget, 600)); f.write(x[b].tobytes()); f.write(init[b].tobytes())
