"""Stand-in for the third-party ``entmax`` package on machines that do not have it, so that
``from entmax import sparsemax, entmax15, SparsemaxLoss, Entmax15Loss`` in the reference's
experiment scripts (experiments/semi_supervised-vae/train.py:9) resolves.  ``sparsemax``,
``entmax15`` and the two Fenchel-Young losses run on the gfx950 kernels of this repository
(``smlvm_sparsemax_*``, ``smlvm_entmax15_*``); the published algorithms they follow are restated
in ``oracle/sparsemax_oracle.c`` and ``oracle/entmax15_oracle.c``."""
import torch
import torch.nn as nn

from lvmhelpers.functional import entmax15, sparsemax  # noqa: F401


class _SparsemaxLossFn(torch.autograd.Function):
    """Fenchel-Young loss of sparsemax: (1 - |p|^2)/2 + <p - e_y, x>; gradient p - e_y."""

    @staticmethod
    def forward(ctx, X, target):
        p = sparsemax(X.detach(), dim=-1)
        loss = (1 - (p * p).sum(dim=1)) / 2
        p = p.clone()
        p.scatter_add_(1, target.unsqueeze(1), torch.full_like(p, -1))
        loss = loss + torch.einsum("ij,ij->i", p, X)
        ctx.save_for_backward(p)
        return loss

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return g.unsqueeze(1) * p, None


class _Entmax15LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, target):
        p = entmax15(X.detach(), dim=-1)
        loss = (1 - (p * torch.sqrt(p)).sum(dim=1)) / 0.75
        p = p.clone()
        p.scatter_add_(1, target.unsqueeze(1), torch.full_like(p, -1))
        loss = loss + torch.einsum("ij,ij->i", p, X)
        ctx.save_for_backward(p)
        return loss

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return g.unsqueeze(1) * p, None


class _FYLoss(nn.Module):
    _fn = None

    def __init__(self, k=None, ignore_index=-100, reduction="elementwise_mean"):
        super().__init__()
        assert reduction in ("elementwise_mean", "sum", "none")
        self.reduction = reduction
        self.ignore_index = ignore_index

    def forward(self, X, target):
        loss = self._fn.apply(X, target.clamp(min=0))
        if self.ignore_index >= 0:
            ignored = target == self.ignore_index
            loss = loss.masked_fill(ignored, 0.0)
            size = float((target.size(0) - ignored.sum()).item())
        else:
            size = float(target.size(0))
        if self.reduction == "sum":
            return loss.sum()
        if self.reduction == "elementwise_mean":
            return loss.sum() / size
        return loss


class SparsemaxLoss(_FYLoss):
    _fn = _SparsemaxLossFn


class Entmax15Loss(_FYLoss):
    _fn = _Entmax15LossFn
