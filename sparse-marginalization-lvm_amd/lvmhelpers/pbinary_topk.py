"""k-best bit vectors with the reference's host-array API (lvmhelpers/pbinary_topk.pyx:18-34).

The arrays live on the host (that is the reference's calling convention), so these two entry
points copy to the GPU, run the batched kernel and copy back; ``structmarg.TopKSparsemaxWrapper``
calls the kernel directly on device tensors instead."""
import numpy as np
import torch

from . import _native


def _device():
    if not torch.cuda.is_available():
        raise _native.NativeError("pbinary_topk needs a GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def binary_topk(scores, k):
    """scores: sequence of floats -> list of (score, [0/1]*n), best first (pbinary_topk.pyx:18-20)."""
    x = torch.tensor([float(v) for v in scores], dtype=torch.float32, device=_device()).unsqueeze(0)
    cfg, top = _native.binary_topk_batched(x, int(k), want_scores=True)
    cfg = cfg[0].to(torch.int64).cpu().tolist()
    top = top[0].cpu().tolist()
    return [(s, c) for s, c in zip(top, cfg)]


def batched_topk(scores, configs, k):
    """scores float32 [B,n], configs float32 [B,k,n] (filled in place), as pbinary_topk.pyx:23-34."""
    for name, a, nd in (("scores", scores, 2), ("configs", configs, 3)):
        if not isinstance(a, np.ndarray) or a.dtype != np.float32:
            raise ValueError(f"Buffer dtype mismatch, expected 'float' for {name}")
        if a.ndim != nd:
            raise ValueError(f"Buffer has wrong number of dimensions (expected {nd}, got {a.ndim})")
    B, n = scores.shape
    if configs.shape != (B, k, n):
        raise ValueError("configs must be [B, k, n]")
    x = torch.from_numpy(np.ascontiguousarray(scores)).to(_device())
    cfg, _ = _native.binary_topk_batched(x, int(k))
    configs[...] = cfg.cpu().numpy()
