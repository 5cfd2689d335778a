import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'sparse-marginalization-lvm_amd'))
import numpy as np, torch, oracle
from lvmhelpers import _native
for K, sigma in ((4096, 0.1), (4096, 0.05), (2048, 0.02), (4096, 0.005), (3000, 0.05)):
    R = 256
    rng = np.random.default_rng(int(1e4 * sigma) + K)
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    _, s0, tau0, _ = oracle.sparsemax_fwd(x)
    mx = x.max(axis=1)
    for r in range(0, R, 2):
        t = np.float32(tau0[r] + mx[r]); below = np.flatnonzero(x[r] < t - 2e-3); rng.shuffle(below)
        n = min(below.size, 400); k = r % 8
        if k == 0: x[r, below[:n]] = t - np.float32(5e-4)
        elif k == 2:
            x[r, below[:2]] = np.nextafter(t, np.float32(-np.inf)); x[r, below[2:n]] = t - rng.uniform(1e-5, 2e-3, n - 2).astype(np.float32)
        elif k == 4:
            x[r, below[:2]] = t; x[r, below[2:n]] = t - np.float32(1.5e-3)
        else: x[r, below[:n]] = t - rng.uniform(1e-5, 2e-3, n).astype(np.float32)
    p_ref, s_ref, tau_ref, _ = oracle.sparsemax_fwd(x)
    p, supp, _, tau, _, cols = _native.sparsemax_fwd(torch.from_numpy(x).cuda(), want_tau=True, want_cols=True)
    supp = supp.cpu().numpy(); tau = tau.cpu().numpy(); p = p.cpu().numpy()
    bad = np.flatnonzero((supp != s_ref) | (tau != tau_ref) | (p != p_ref).any(1))
    print(K, sigma, 'bad rows', len(bad))
    for r in bad[:12]:
        z = np.sort(x[r] - x[r].max())[::-1].astype(np.float32)
        cs = np.cumsum(z.astype(np.float64)); j = np.arange(1, K + 1).astype(np.float32)
        test = j * z > (cs.astype(np.float32) - np.float32(1))
        print('  row', r, 'k', r % 8, 's0', s0[r], 'ref', s_ref[r], 'got', supp[r], 'tau', tau_ref[r], tau[r],
              'passing beyond s0:', np.flatnonzero(test)[s0[r]:][:8], 'pdiff', np.abs(p[r]-p_ref[r]).max())
