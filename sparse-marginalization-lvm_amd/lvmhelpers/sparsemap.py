"""Per-sample SparseMAP with the reference's surface (lvmhelpers/sparsemap.py): the autograd
functions ``SparseMAP`` / ``BernSparseMAP`` / ``BudgetSparseMAP`` (sparsemap.py:10-44,90-127),
``SequenceSparseMAP`` / ``SequenceBinarySparseMAP`` (sparsemap.py:47-87,130-148) and the entry points
``bernoulli_smap`` / ``budget_smap`` / ``sequence_smap`` (sparsemap.py:151-160).

As in the reference, a function call builds one factor object (``PFactorBernoulli`` /
``PFactorBudget``), keeps it on the autograd ``ctx`` (``ctx.f``) and the backward asks that factor
for the Jacobian-vector product; here the factor is a batch of one sample on the GPU solver and
its state never leaves the device.  ``structmarg.SparseMAPWrapper`` does not go through these: it
solves the whole batch in one launch (``functional.sparsemap_batched``).
"""
import torch

from .pbernoulli import PFactorBernoulli
from .pbudget import PFactorBudget
from .psequence import PFactorSequenceBinary


class SparseMAP(torch.autograd.Function):
    """Shared plumbing of the per-sample functions (reference: sparsemap.py:10-44)."""

    @classmethod
    def run_sparsemap(cls, ctx, x):
        if x.dim() != 1:
            raise ValueError("expected a 1-D score vector")
        ctx.n = x.shape[0]
        ctx.f = cls.make_factor(ctx)
        # same draw as the reference: torch.rand(2n, dtype=double) from the global CPU generator,
        # handed to init_active_set_from_scores (sparsemap.py:25-27)
        init = torch.rand(2 * ctx.n, dtype=torch.double) if ctx.init else None
        p, aset = ctx.f._solve_device(x.detach().to(torch.float32).contiguous(), init, ctx.max_iter)
        ctx.mark_non_differentiable(aset)
        return p, aset

    @classmethod
    def jv(cls, ctx, dp):
        return ctx.f._jacobian_vec_device(dp.contiguous())


class SequenceSparseMAP(torch.autograd.Function):
    """Plumbing of the sequence factors (reference: sparsemap.py:47-87)."""

    @classmethod
    def run_sparsemap(cls, ctx, x, t):
        if x.dim() != 1 or t.dim() != 1 or t.shape[0] != x.shape[0] - 1:
            raise ValueError("expected x [n] and t [n-1]")
        ctx.n = x.shape[0]
        ctx.f = cls.make_factor(ctx)
        # torch.rand(2n, dtype=double) from the global CPU generator; the additional potentials of
        # the init call are zeros (sparsemap.py:66-69)
        init = torch.rand(2 * ctx.n, dtype=torch.double) if ctx.init else None
        p, aset = ctx.f._solve_device(x.detach().to(torch.float32).contiguous(),
                                      t.detach().to(torch.float32).contiguous(), init, ctx.max_iter)
        ctx.mark_non_differentiable(aset)
        return p, aset

    @classmethod
    def jv(cls, ctx, dp):
        return ctx.f._jacobian_vec_device(dp.contiguous())


class BernSparseMAP(SparseMAP):
    """p, aset = BernSparseMAP.apply(x, max_iter, init)   (reference: sparsemap.py:90-106)."""

    @classmethod
    def make_factor(cls, ctx):
        f = PFactorBernoulli()
        f.initialize(ctx.n)
        return f

    @classmethod
    def forward(cls, ctx, x, max_iter, init):
        ctx.max_iter = max_iter
        ctx.init = init
        return cls.run_sparsemap(ctx, x)

    @classmethod
    def backward(cls, ctx, dp, daset):
        return cls.jv(ctx, dp), None, None


class BudgetSparseMAP(SparseMAP):
    """p, aset = BudgetSparseMAP.apply(x, budget, max_iter, init)   (reference: sparsemap.py:109-127)."""

    @classmethod
    def make_factor(cls, ctx):
        f = PFactorBudget()
        f.initialize(ctx.n, ctx.budget)
        return f

    @classmethod
    def forward(cls, ctx, x, budget, max_iter, init):
        ctx.init = init
        ctx.max_iter = max_iter
        ctx.budget = budget
        return cls.run_sparsemap(ctx, x)

    @classmethod
    def backward(cls, ctx, dp, daset):
        return cls.jv(ctx, dp), None, None, None


class SequenceBinarySparseMAP(SequenceSparseMAP):
    """p, aset = SequenceBinarySparseMAP.apply(x, t, max_iter, init)   (reference: sparsemap.py:130-148)."""

    @classmethod
    def make_factor(cls, ctx):
        f = PFactorSequenceBinary()
        f.initialize(ctx.n)
        return f

    @classmethod
    def forward(cls, ctx, x, t, max_iter, init):
        ctx.init = init
        ctx.max_iter = max_iter
        return cls.run_sparsemap(ctx, x, t)

    @classmethod
    def backward(cls, ctx, dp, daset):
        d_eta_u, d_eta_v = cls.jv(ctx, dp)
        return d_eta_u, d_eta_v, None, None


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
