"""PFactorBernoulli with the methods the reference calls on it (lvmhelpers/pbernoulli.pyx:10-22 plus
the lpsmap PGenericFactor methods used at lvmhelpers/sparsemap.py:27-30,42).  One factor = a batch
of one sample on the GPU solver."""
import numpy as np
import torch

from . import _native


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

    def init_active_set_from_scores(self, eta_u, eta_v):
        u = np.asarray(eta_u, dtype=np.float64).reshape(1, -1)
        if u.shape[1] != 2 * self.n:
            raise ValueError("expected 2n scores")
        self._init = torch.from_numpy(u).to(self._dev())

    def solve_qp(self, eta_u, eta_v, max_iter=10):
        u = np.asarray(eta_u, dtype=np.float64)
        if u.shape[0] != 2 * self.n or np.any(u[self.n:] != 0) \
                or np.any(u[:self.n].astype(np.float32).astype(np.float64) != u[:self.n]):
            raise NotImplementedError(
                "the GPU factor takes eta = [float32 scores; 0] as lvmhelpers/sparsemap.py:21-22 builds it")
        x = torch.from_numpy(u[:self.n].astype(np.float32)).unsqueeze(0).to(self._dev())
        out = _native.sparsemap_fwd(x, self._init, self._budget, max_iter)
        self._state = out
        m = int(out["alen"][0].item())
        aset, p = self.get_sparse_solution()
        post = np.zeros(2 * self.n)
        for pi, cfg in zip(p[:m], aset):
            c = np.asarray(cfg)
            post[:self.n] += pi * c
            post[self.n:] += pi * (1 - c)
        return post, np.zeros(0)

    def get_sparse_solution(self):
        out = self._state
        m = int(out["alen"][0].item())
        rows = torch.zeros(m, dtype=torch.int32, device=out["p"].device)
        cols = torch.arange(m, dtype=torch.int32, device=out["p"].device)
        aset = _native.unpack_configs(out["aset_bits"], rows, cols, self.n)
        return aset.to(torch.int64).cpu().tolist(), out["p"][0, :m].double().cpu().tolist()

    def dist_jacobian_vec(self, dp, out_u, out_v):
        out = self._state
        cap = self.n + 1
        g = torch.zeros((1, cap), dtype=torch.float32, device=out["p"].device)
        dp = np.asarray(dp, dtype=np.float32)
        g[0, :dp.shape[0]] = torch.from_numpy(dp).to(g.device)
        d = _native.sparsemap_bwd(g, out["aset_bits"], out["alen"], out["inv_a"], self.n)
        out_u[:self.n] = d[0].cpu().numpy()
        out_u[self.n:] = 0  # the reference only reads the first n entries (sparsemap.py:43)
