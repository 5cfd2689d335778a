"""Batch data-parallelism for the hot path: one process per GPU, torch.distributed ("nccl" is RCCL
on ROCm, over xGMI inside a node).

The path shards along the batch/row axis only: every sparsemax row and every SparseMAP / top-k
sample is independent, so the forward/backward kernels need NO collective.  Training adds exactly
one exchange per step: the gradients of all parameters are flattened into a single fp32 bucket
(2.3-8.9 MB for the reference's three models, i.e. latency-bound on 153 GB/s xGMI links: one
bucket, no overlap machinery), all-reduced (sum) and divided by the world size.  The reference
normalises every loss by the LOCAL batch size (lvmhelpers/marg.py:126, structmarg.py:140,242), so
equal per-rank batches + gradient averaging reproduce the global-batch gradient.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` rows owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(tensor, rank=None, world=None, dim=0):
    """The rank's contiguous slice of `tensor` along `dim`."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(tensor.shape[dim], rank, world)
    return tensor.narrow(dim, lo, hi - lo)


def global_init_scores(batch_total, n, rank, world, device, seed=None):
    """SparseMAP init scores for the rank's shard of a GLOBAL batch: every rank draws the whole
    [batch_total, 2n] fp64 block from the same CPU stream (as lvmhelpers/sparsemap.py:26 would for
    the unsharded batch; seed=None: torch's global CPU generator, which equal seeds keep in step
    on all ranks) and keeps its rows, so results do not depend on the world size."""
    g = None
    if seed is not None:
        g = torch.Generator().manual_seed(seed)
    full = torch.rand(batch_total, 2 * n, dtype=torch.double, generator=g)
    lo, hi = shard_range(batch_total, rank, world)
    return full[lo:hi].to(device)


def current_shard(local_batch):
    """(rank, world, batch_total) of this process when torch.distributed runs with more than one
    rank and equal shards, else None."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        return dist.get_rank(), world, local_batch * world
    return None


@torch.no_grad()
def allreduce_gradients(parameters, group=None, local_rows=None):
    """Global-batch gradient from the ranks' local-batch gradients with ONE all-reduce of a flat
    fp32 bucket.

    The bucket covers EVERY parameter that requires grad, in the same order on every rank; a
    parameter without a gradient on this rank (a decoder branch no row of the shard used)
    contributes zeros and gets the reduced gradient back, so bucket lengths never differ between
    ranks.  ``local_rows``: this rank's batch size.  The reference normalises a loss by the LOCAL
    batch (lvmhelpers/marg.py:126, structmarg.py:140,242), so the global-batch gradient is
    sum_r rows_r * grad_r / sum_r rows_r: the row counts ride along in the same bucket.  Without
    it the shards are taken to be equal (plain mean over ranks).  Returns the bytes exchanged.
    """
    params = [p for p in parameters if p.requires_grad]
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or not params:
        return 0
    dev = params[0].device
    weight = 1.0 if local_rows is None else float(local_rows)
    parts = [(p.grad.reshape(-1).to(torch.float32) if p.grad is not None
              else torch.zeros(p.numel(), dtype=torch.float32, device=dev)) for p in params]
    parts.append(torch.full((1,), weight, dtype=torch.float32, device=dev))
    flat = torch.cat(parts)
    if local_rows is not None:
        flat[:-1].mul_(weight)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat[:-1].div_(flat[-1])     # sum of the weights: the world size, or the global row count
    off = 0
    for p in params:
        k = p.numel()
        g = flat[off:off + k].view_as(p)
        if p.grad is None:
            p.grad = g.to(p.dtype).clone()
        else:
            p.grad.copy_(g)
        off += k
    return flat.numel() * 4


def allreduce_mean_scalar(value, group=None):
    """Mean of a python/0-d scalar over ranks (for reporting log values only)."""
    t = torch.as_tensor(value, dtype=torch.float64).reshape(1).clone()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=group)
        t /= dist.get_world_size(group)
    return float(t.item())
