"""PFactorBernoulli with the methods the reference calls on it (lvmhelpers/pbernoulli.pyx:10-22 plus
the lpsmap PGenericFactor methods used at lvmhelpers/sparsemap.py:27-30,42): initialize,
init_active_set_from_scores, solve_qp, get_sparse_solution, dist_jacobian_vec.  One factor = a
batch of one sample on the GPU solver (csrc/sparsemap.hip); the state the reference's factor
object keeps between forward and backward (active set + inverse KKT matrix) stays on the device.
"""
import numpy as np
import torch

from . import _native


def _as_f64(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64).reshape(-1)


class PFactorBernoulli:
    _budget = 0

    def __init__(self, allocate=True):
        self.n = None
        self._init = None
        self._state = None

    def initialize(self, length):
        self.n = int(length)

    def _dev(self):
        if not torch.cuda.is_available():
            raise _native.NativeError("the SparseMAP factor needs a GPU (no CPU fallback)")
        return torch.device("cuda", torch.cuda.current_device())

    # ------------------------------------------------------------------ device-side core
    def _solve_device(self, scores, init_scores, max_iter):
        """scores [n] f32 on the GPU, init_scores [2n] f64 (any device) or None.  Returns
        (p [|A|] f32 differentiable-by-jv, aset [|A|, n] f32 0/1), both on the GPU."""
        n = self.n
        init = None if init_scores is None else init_scores.reshape(1, 2 * n).to(scores.device)
        out = _native.sparsemap_fwd(scores.reshape(1, n), init, self._budget, max_iter)
        self._state = out
        m = int(out["alen"][0].item())
        rows = torch.zeros(m, dtype=torch.int32, device=scores.device)
        cols = torch.arange(m, dtype=torch.int32, device=scores.device)
        aset = _native.unpack_configs(out["aset_bits"], rows, cols, n)
        return out["p"][0, :m], aset

    def _jacobian_vec_device(self, dp):
        """dp [|A|] f32 on the GPU -> d eta_u[:n] f32 on the GPU (sparsemap.py:38-44)."""
        out = self._state
        g = torch.zeros((1, self.n + 1), dtype=torch.float32, device=out["p"].device)
        g[0, :dp.shape[0]] = dp.to(torch.float32)
        return _native.sparsemap_bwd(g, out["aset_bits"], out["alen"], out["inv_a"], self.n)[0]

    # ------------------------------------------------------------------ the reference's host API
    def init_active_set_from_scores(self, eta_u, eta_v):
        u = _as_f64(eta_u)
        if u.shape[0] != 2 * self.n:
            raise ValueError("expected 2n scores")
        if len(eta_v) != 0:
            raise ValueError("this factor has no additional variables")
        self._init = torch.from_numpy(u.copy())

    def solve_qp(self, eta_u, eta_v, max_iter=10):
        """min_p 1/2 |M p - eta|^2 over the simplex; columns of M are [y; 1 - y].  The minimiser
        only depends on e = eta_u[:n] - eta_u[n:], which is what the kernel takes as fp32 scores
        (lvmhelpers/sparsemap.py:21-22 always passes [float32 scores; 0]); e must be exactly
        representable in fp32.  Returns (posteriors [2n], additional posteriors [0])."""
        u = _as_f64(eta_u)
        if u.shape[0] != 2 * self.n:
            raise ValueError("expected 2n potentials")
        if len(eta_v) != 0:
            raise ValueError("this factor has no additional variables")
        e = u[:self.n] - u[self.n:]
        if np.any(e.astype(np.float32).astype(np.float64) != e):
            raise NotImplementedError(
                "the GPU factor takes eta_u[:n] - eta_u[n:] as fp32 scores; these potentials are not "
                "representable (lvmhelpers/sparsemap.py:21-22 builds eta = [float32 scores; 0])")
        x = torch.from_numpy(e.astype(np.float32)).to(self._dev())
        p, aset = self._solve_device(x, self._init, max_iter)
        mu = (p.double().unsqueeze(1) * aset.double()).sum(0).cpu().numpy()
        total = float(p.double().sum().item())
        return np.concatenate([mu, total - mu]), np.zeros(0)

    def get_sparse_solution(self):
        out = self._state
        m = int(out["alen"][0].item())
        rows = torch.zeros(m, dtype=torch.int32, device=out["p"].device)
        cols = torch.arange(m, dtype=torch.int32, device=out["p"].device)
        aset = _native.unpack_configs(out["aset_bits"], rows, cols, self.n)
        return aset.to(torch.int64).cpu().tolist(), out["p"][0, :m].double().cpu().tolist()

    def dist_jacobian_vec(self, dp, out_u, out_v):
        dp = torch.from_numpy(np.asarray(dp, dtype=np.float32).reshape(-1)).to(self._state["p"].device)
        d = self._jacobian_vec_device(dp)
        out_u[:self.n] = d.cpu().numpy()
        out_u[self.n:] = 0  # the reference only reads the first n entries (sparsemap.py:43)
