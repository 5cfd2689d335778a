"""Per-sample SparseMAP with the reference's surface (lvmhelpers/sparsemap.py): the autograd
functions ``SparseMAP`` / ``BernSparseMAP`` / ``BudgetSparseMAP`` (sparsemap.py:10-44,90-127),
``SequenceSparseMAP`` / ``SequenceBinarySparseMAP`` (sparsemap.py:47-87,130-148) and the entry points
``bernoulli_smap`` / ``budget_smap`` / ``sequence_smap`` (sparsemap.py:151-160).

As in the reference, a call builds one factor object (``PFactorBernoulli`` / ``PFactorBudget`` /
``PFactorSequenceBinary``), keeps it on the autograd ``ctx`` (``ctx.f``) and the backward asks that
factor for the Jacobian-vector product; here the factor is a batch of one sample on the GPU solver
and its state never leaves the device.  ``structmarg.SparseMAPWrapper`` does not go through these:
it solves the whole batch in one launch (``functional.sparsemap_batched``).

All five classes share one piece of plumbing (``_solve`` / ``_pull_back`` below); what differs is
the factor they build and how many potential tensors they take (scores, or scores + edge
potentials).
"""
import torch

from .pbernoulli import PFactorBernoulli
from .pbudget import PFactorBudget
from .psequence import PFactorSequenceBinary


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _solve(cls, ctx, potentials):
    """Build the factor, draw the init scores like the reference does -- torch.rand(2n,
    dtype=double) from the global CPU generator, sparsemap.py:25-27,66-69 -- and solve on the GPU."""
    x = potentials[0]
    if x.dim() != 1:
        raise ValueError("expected a 1-D score vector")
    ctx.n = x.shape[0]
    ctx.f = cls.make_factor(ctx)
    init = torch.rand(2 * ctx.n, dtype=torch.double) if ctx.init else None
    p, aset = ctx.f._solve_device(*[_f32(t) for t in potentials], init, ctx.max_iter)
    ctx.mark_non_differentiable(aset)
    return p, aset


class SparseMAP(torch.autograd.Function):
    """Plumbing of the bit-vector functions (reference: sparsemap.py:10-44)."""

    @classmethod
    def run_sparsemap(cls, ctx, x):
        return _solve(cls, ctx, (x,))

    @classmethod
    def jv(cls, ctx, dp):
        return ctx.f._jacobian_vec_device(dp.contiguous())


class SequenceSparseMAP(torch.autograd.Function):
    """Plumbing of the sequence functions (reference: sparsemap.py:47-87): node scores x [n] and
    additional (edge) potentials t [n-1]; the Jacobian-vector product has both parts."""

    @classmethod
    def run_sparsemap(cls, ctx, x, t):
        if t.dim() != 1 or x.dim() != 1 or t.shape[0] != x.shape[0] - 1:
            raise ValueError("expected x [n] and t [n-1]")
        return _solve(cls, ctx, (x, t))

    @classmethod
    def jv(cls, ctx, dp):
        return ctx.f._jacobian_vec_device(dp.contiguous())   # (d eta_u[:n], d eta_v)


class BernSparseMAP(SparseMAP):
    """p, aset = BernSparseMAP.apply(x, max_iter, init)   (reference: sparsemap.py:90-106)."""

    @classmethod
    def make_factor(cls, ctx):
        factor = PFactorBernoulli()
        factor.initialize(ctx.n)
        return factor

    @classmethod
    def forward(cls, ctx, x, max_iter, init):
        ctx.max_iter, ctx.init = max_iter, init
        return cls.run_sparsemap(ctx, x)

    @classmethod
    def backward(cls, ctx, dp, daset):
        return cls.jv(ctx, dp), None, None


class BudgetSparseMAP(SparseMAP):
    """p, aset = BudgetSparseMAP.apply(x, budget, max_iter, init)   (reference: sparsemap.py:109-127)."""

    @classmethod
    def make_factor(cls, ctx):
        factor = PFactorBudget()
        factor.initialize(ctx.n, ctx.budget)
        return factor

    @classmethod
    def forward(cls, ctx, x, budget, max_iter, init):
        ctx.budget, ctx.max_iter, ctx.init = budget, max_iter, init
        return cls.run_sparsemap(ctx, x)

    @classmethod
    def backward(cls, ctx, dp, daset):
        return cls.jv(ctx, dp), None, None, None


class SequenceBinarySparseMAP(SequenceSparseMAP):
    """p, aset = SequenceBinarySparseMAP.apply(x, t, max_iter, init)   (reference: sparsemap.py:130-148)."""

    @classmethod
    def make_factor(cls, ctx):
        factor = PFactorSequenceBinary()
        factor.initialize(ctx.n)
        return factor

    @classmethod
    def forward(cls, ctx, x, t, max_iter, init):
        ctx.max_iter, ctx.init = max_iter, init
        return cls.run_sparsemap(ctx, x, t)

    @classmethod
    def backward(cls, ctx, dp, daset):
        return (*cls.jv(ctx, dp), None, None)


def bernoulli_smap(x, max_iter=100, init=True):
    """(p [|A|], aset [|A|, n]) for one score vector x [n] (sparsemap.py:151-152)."""
    return BernSparseMAP.apply(x, max_iter, init)


def budget_smap(x, budget=5, max_iter=100, init=True):
    """Same with at most `budget` bits on (sparsemap.py:155-156)."""
    return BudgetSparseMAP.apply(x, budget, max_iter, init)


def sequence_smap(x, t, max_iter=100, init=True):
    """(p [|A|], aset [|A|, n]) for a binary chain: node scores x [n], edge potentials t [n-1] of
    the (1 -> 1) transitions; differentiable in both (sparsemap.py:159-160)."""
    return SequenceBinarySparseMAP.apply(x, t, max_iter, init)
