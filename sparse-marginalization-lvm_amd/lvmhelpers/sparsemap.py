"""Per-sample SparseMAP entry points with the reference's signatures
(lvmhelpers/sparsemap.py:151-160).  They call the batched kernel with a batch of one; the batched
path used by ``structmarg.SparseMAPWrapper`` is ``functional.sparsemap_batched``."""
import torch

from . import _native
from . import functional as F_


def _single(x, budget, max_iter, init):
    if x.dim() != 1:
        raise ValueError("expected a 1-D score vector")
    n = x.shape[0]
    init_scores = None
    if init:  # same draw as the reference: torch.rand(2n, dtype=double), global CPU generator
        init_scores = torch.rand(2 * n, dtype=torch.double).unsqueeze(0).to(x.device)
    p, aset_bits, alen, _, _ = F_.sparsemap_batched(x.unsqueeze(0), init_scores, budget, max_iter)
    m = int(alen[0].item())
    rows = torch.zeros(m, dtype=torch.int32, device=x.device)
    cols = torch.arange(m, dtype=torch.int32, device=x.device)
    aset = _native.unpack_configs(aset_bits, rows, cols, n)
    return p[0, :m], aset


def bernoulli_smap(x, max_iter=100, init=True):
    """(p [|A|], aset [|A|, n]) for one score vector x [n] (sparsemap.py:90-106,151-152)."""
    return _single(x, 0, max_iter, init)


def budget_smap(x, budget=5, max_iter=100, init=True):
    """Same with at most `budget` bits on (sparsemap.py:109-127,155-156)."""
    return _single(x, budget, max_iter, init)


def sequence_smap(x, t, max_iter=100, init=True):
    """Sequence factor (sparsemap.py:159-160): no experiment uses it; outside this hot path."""
    raise NotImplementedError(
        "sequence_smap is not part of the MI355X hot path (unused by every reference experiment)")
