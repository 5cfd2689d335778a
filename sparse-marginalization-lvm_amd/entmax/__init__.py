"""Stand-in for the third-party ``entmax`` package on machines that do not have it, so that
``from entmax import sparsemax, entmax15, SparsemaxLoss, Entmax15Loss`` in the reference's
experiment scripts (experiments/semi_supervised-vae/train.py:9) resolves.  ``sparsemax`` and
``SparsemaxLoss`` are the gfx950 kernels of this repository; ``entmax15`` / ``Entmax15Loss`` are
plain PyTorch restatements of the published algorithm (Peters et al. 2019), kept only for API
completeness (no experiment script selects them: lvmhelpers/utils.py:36-39)."""
import torch
import torch.nn as nn

from lvmhelpers.functional import sparsemax  # noqa: F401


def _entmax15_fwd(X, dim):
    X = X / 2
    X = X - X.max(dim=dim, keepdim=True)[0]
    Xs, _ = torch.sort(X, dim=dim, descending=True)
    d = X.shape[dim]
    shape = [1] * X.dim()
    shape[dim] = -1
    rho = torch.arange(1, d + 1, device=X.device, dtype=X.dtype).view(shape)
    mean = Xs.cumsum(dim) / rho
    mean_sq = (Xs ** 2).cumsum(dim) / rho
    ss = rho * (mean_sq - mean ** 2)
    delta = (1 - ss) / rho
    tau = mean - torch.sqrt(torch.clamp(delta, 0))
    support = (tau <= Xs).sum(dim=dim, keepdim=True)
    tau_star = tau.gather(dim, support - 1)
    return torch.clamp(X - tau_star, min=0) ** 2


class _Entmax15(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, dim):
        Y = _entmax15_fwd(X, dim)
        ctx.dim = dim
        ctx.save_for_backward(Y)
        return Y

    @staticmethod
    def backward(ctx, dY):
        (Y,) = ctx.saved_tensors
        gppr = Y.sqrt()
        dX = dY * gppr
        q = dX.sum(ctx.dim, keepdim=True) / gppr.sum(ctx.dim, keepdim=True)
        return dX - q * gppr, None


def entmax15(X, dim=-1):
    return _Entmax15.apply(X, dim)


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
        p = _entmax15_fwd(X.detach(), -1)
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
