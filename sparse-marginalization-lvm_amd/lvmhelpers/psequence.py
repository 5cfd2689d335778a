"""PFactorSequenceBinary with the methods the reference calls on it (lvmhelpers/psequence.pyx:25-37
plus the lpsmap PGenericFactor methods used at lvmhelpers/sparsemap.py:63,69,71-72,84):
initialize, set_additional_log_potentials, init_active_set_from_scores, solve_qp,
get_sparse_solution, dist_jacobian_vec.  One factor = a batch of one chain on the GPU solver
(csrc/sparsemap.hip, SEQ = true: Viterbi oracle of lvmhelpers/sequence_binary.h:11-67 on
sequence.h:66-166); active set and inverse KKT matrix stay on the device between forward and
backward.  The general multi-state ``PFactorSequence`` (psequence.pyx:10-22) is never constructed
by the reference's Python code and is not provided.
"""
import numpy as np
import torch

from . import _native
from .pbernoulli import PFactorBernoulli, _as_f64


class PFactorSequenceBinary(PFactorBernoulli):

    def __init__(self, allocate=True):
        super().__init__(allocate)
        self._trans = None

    def initialize(self, length):
        length = int(length)
        if length < 2:
            raise ValueError("a sequence factor needs at least two positions")
        self.n = length

    def set_additional_log_potentials(self, additionals):
        t = _as_f64(additionals)
        if t.shape[0] != self.n - 1:
            raise ValueError("expected n-1 edge potentials")
        self._trans = t

    # ------------------------------------------------------------------ device-side core
    def _solve_device(self, scores, trans, init_scores, max_iter):
        """scores [n], trans [n-1] f32 on the GPU, init_scores [2n] f64 or None ->
        (p [|A|] f32, aset [|A|, n] f32 0/1) on the GPU."""
        n = self.n
        init = None if init_scores is None else init_scores.reshape(1, 2 * n).to(scores.device)
        out = _native.sparsemap_seq_fwd(scores.reshape(1, n), trans.reshape(1, n - 1), init, max_iter)
        self._state = out
        m = int(out["alen"][0].item())
        rows = torch.zeros(m, dtype=torch.int32, device=scores.device)
        cols = torch.arange(m, dtype=torch.int32, device=scores.device)
        aset = _native.unpack_configs(out["aset_bits"], rows, cols, n)
        return out["p"][0, :m], aset

    def _jacobian_vec_device(self, dp):
        """dp [|A|] f32 on the GPU -> (d eta_u[:n], d eta_v) f32 on the GPU (sparsemap.py:79-87)."""
        out = self._state
        g = torch.zeros((1, self.n + 1), dtype=torch.float32, device=out["p"].device)
        g[0, :dp.shape[0]] = dp.to(torch.float32)
        du, dv = _native.sparsemap_seq_bwd(g, out["aset_bits"], out["alen"], out["inv_a"], self.n)
        return du[0], dv[0]

    # ------------------------------------------------------------------ the reference's host API
    def init_active_set_from_scores(self, eta_u, eta_v):
        u = _as_f64(eta_u)
        if u.shape[0] != 2 * self.n:
            raise ValueError("expected 2n scores")
        v = _as_f64(eta_v)
        if v.shape[0] not in (0, self.n - 1) or np.any(v != 0):
            raise NotImplementedError(
                "the GPU factor initialises with zero edge potentials, as the reference does "
                "(lvmhelpers/sparsemap.py:67-69)")
        self._init = torch.from_numpy(u.copy())

    def solve_qp(self, eta_u, eta_v, max_iter=10):
        """Returns (posteriors [2n], additional posteriors [n-1]); eta_u[:n] - eta_u[n:] and eta_v
        must be representable in fp32 (sparsemap.py:58-62 builds them from float32 tensors)."""
        u = _as_f64(eta_u)
        v = _as_f64(eta_v)
        if u.shape[0] != 2 * self.n or v.shape[0] != self.n - 1:
            raise ValueError("expected 2n variable and n-1 additional potentials")
        e = u[:self.n] - u[self.n:]
        if np.any(e.astype(np.float32).astype(np.float64) != e) or \
           np.any(v.astype(np.float32).astype(np.float64) != v):
            raise NotImplementedError("the GPU factor takes fp32 scores and edge potentials")
        dev = self._dev()
        x = torch.from_numpy(e.astype(np.float32)).to(dev)
        t = torch.from_numpy(v.astype(np.float32)).to(dev)
        p, aset = self._solve_device(x, t, self._init, max_iter)
        pd, ad = p.double(), aset.double()
        mu = (pd.unsqueeze(1) * ad).sum(0).cpu().numpy()
        nu = (pd.unsqueeze(1) * ad[:, :-1] * ad[:, 1:]).sum(0).cpu().numpy()
        total = float(pd.sum().item())
        return np.concatenate([mu, total - mu]), nu

    def dist_jacobian_vec(self, dp, out_u, out_v):
        dp = torch.from_numpy(np.asarray(dp, dtype=np.float32).reshape(-1)).to(self._state["p"].device)
        du, dv = self._jacobian_vec_device(dp)
        out_u[:self.n] = du.cpu().numpy()
        out_u[self.n:] = 0  # the reference only reads the first n entries (sparsemap.py:85)
        out_v[:] = dv.cpu().numpy()
