"""torch.autograd glue over the C-ABI kernels (``_native``).  Host-side plumbing only."""
import torch

from . import _native


def _patterns_written(x, p, mask):
    """Did this forward call leave every row's 16-bit-per-lane pattern in `mask`?  Only the vector
    path of the 16-register tiles does (32 < K <= 1024, K % 4 == 0, x and p 16-byte aligned); the
    pattern-fed compaction must not be handed a buffer that was never filled."""
    K = x.shape[-1]
    return (mask is not None and 32 < K <= 1024 and K % 4 == 0
            and x.data_ptr() % 16 == 0 and p.data_ptr() % 16 == 0)


class _Sparsemax(torch.autograd.Function):
    """Row-wise sparsemax over the last dim (replaces entmax.sparsemax; marg.py:45,51)."""

    @staticmethod
    def forward(ctx, x):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        p, supp, _, _, _, cols, mask = _native.sparsemax_fwd(x2, want_cols=True, want_mask=True)
        ctx.has_mask = mask is not None
        ctx.save_for_backward(p, supp, cols, *([mask] if mask is not None else []))
        ctx.mark_non_differentiable(cols)
        patterns = mask if _patterns_written(x2, p, mask) else None
        if patterns is not None:
            ctx.mark_non_differentiable(patterns)
        return p.view(shape), cols, patterns

    @staticmethod
    def backward(ctx, dy, _dcols=None, _dpatterns=None):
        p, supp, cols = ctx.saved_tensors[:3]
        mask = ctx.saved_tensors[3] if ctx.has_mask else None
        dx = _native.sparsemax_bwd_cols(dy.reshape(p.shape).contiguous(), cols, p, supp, mask)
        return dx.view(dy.shape)


class _Entmax15(torch.autograd.Function):
    """Row-wise 1.5-entmax over the last dim (replaces entmax.entmax15; marg.py:46)."""

    @staticmethod
    def forward(ctx, x):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        ctx.listed = _native.entmax15_has_lists(shape[-1]) and x2.shape[0] > 0
        if ctx.listed:   # rows of one wave: the forward leaves each row's support list and sqrt(p)
            p, cols, vals = _native.entmax15_fwd_lists(x2)
            ctx.save_for_backward(p, cols, vals)
        else:
            p, _, _ = _native.entmax15_fwd(x2)
            ctx.save_for_backward(p)
        return p.view(shape)

    @staticmethod
    def backward(ctx, dy):
        p = ctx.saved_tensors[0]
        dy2 = dy.reshape(p.shape).contiguous()
        if ctx.listed:
            return _native.entmax15_bwd_lists(dy2, p, ctx.saved_tensors[1], ctx.saved_tensors[2]).view(dy.shape)
        return _native.entmax15_bwd(dy2, p).view(dy.shape)


class _SparsemaxEntropy(torch.autograd.Function):
    """Fused sparsemax + Shannon entropy (marg.py:51-52): one forward kernel, one backward kernel."""

    @staticmethod
    def forward(ctx, x):
        p, supp, _, _, ent, cols, mask = _native.sparsemax_fwd(x, want_entropy=True, want_cols=True,
                                                               want_mask=True)
        ctx.has_mask = mask is not None
        ctx.save_for_backward(p, supp, cols, *([mask] if mask is not None else []))
        ctx.mark_non_differentiable(supp, cols)
        patterns = mask if _patterns_written(x, p, mask) else None
        if patterns is not None:
            ctx.mark_non_differentiable(patterns)
        return p, ent, supp, cols, patterns

    @staticmethod
    def backward(ctx, dp, dent, _dsupp, _dcols=None, _dpatterns=None):
        p, supp, cols = ctx.saved_tensors[:3]
        mask = ctx.saved_tensors[3] if ctx.has_mask else None
        if dp is None and dent is None:
            return torch.zeros_like(p)
        if dent is None:
            g = dp
        elif dp is None:
            g = _entropy_vjp(p, dent)
        else:
            g = dp + _entropy_vjp(p, dent)
        return _native.sparsemax_bwd_cols(g.contiguous(), cols, p, supp, mask)


def _entropy_vjp(p, dent):
    # d/dp of -sum_{p>0} pc*log(pc), pc = clamp(p, eps, 1-eps): the clamp passes gradient only
    # inside [eps, 1-eps] (marg.py:18-28)
    eps = torch.finfo(p.dtype).eps
    inside = (p >= eps) & (p <= 1 - eps)
    safe = torch.where(inside, p, torch.ones_like(p))
    return torch.where(inside, -(safe.log() + 1.0), torch.zeros_like(p)) * dent.unsqueeze(-1)


class _Entropy(torch.autograd.Function):
    """lvmhelpers/marg.py:6-28 on an arbitrary probability matrix."""

    @staticmethod
    def forward(ctx, p):
        ctx.save_for_backward(p)
        return _native.entropy(p)

    @staticmethod
    def backward(ctx, dent):
        (p,) = ctx.saved_tensors
        return _entropy_vjp(p, dent)


class _WeightedLossReduce(torch.autograd.Function):
    """inv_b * <vals, comps> with the fused deterministic reduction kernel (marg.py:125-126,
    structmarg.py:140,242)."""

    @staticmethod
    def forward(ctx, vals, comps, inv_b):
        total, _ = _native.weighted_loss_reduce_fwd(vals, comps, None, inv_b)
        ctx.save_for_backward(vals, comps)
        ctx.inv_b = inv_b
        return total.reshape(())

    @staticmethod
    def backward(ctx, g):
        vals, comps = ctx.saved_tensors
        dv, dc = _native.weighted_loss_reduce_bwd(
            vals, comps, g.to(torch.float32), ctx.inv_b,
            want_dvals=ctx.needs_input_grad[0], want_dcomps=ctx.needs_input_grad[1])
        return dv, dc, None


class _SupportEntropy(torch.autograd.Function):
    """-inv_b * sum_{p > 0} p log p in two launches forward and one backward (structmarg.py:51-54,
    200: seven elementwise launches forward, six backward, at batch 64 all of them launch latency)."""

    @staticmethod
    def forward(ctx, p, inv_b):
        p = p.contiguous()
        ctx.save_for_backward(p)
        ctx.inv_b = inv_b
        return _native.support_entropy_fwd(p, inv_b).reshape(())

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return _native.support_entropy_bwd(p, g.to(torch.float32), ctx.inv_b), None


def support_entropy(p, inv_b=1.0):
    """-(p[p > 0] @ log p[p > 0]) * inv_b as a 0-d tensor (no clamping: structmarg.py:51-54, 200)."""
    return _SupportEntropy.apply(p, float(inv_b))


class _GatherSupport(torch.autograd.Function):
    """vals = p[rows, cols] for the compaction's (rows, cols), whose values the compaction kernel has
    already written: the forward launches nothing, the backward is one scatter of distinct
    positions (torch's index backward zero-fills, sorts the indices and accumulates: 7 launches)."""

    @staticmethod
    def forward(ctx, p, row_offsets, cols, vals):
        ctx.save_for_backward(row_offsets, cols)
        ctx.shape = tuple(p.shape)
        return vals.view_as(vals)

    @staticmethod
    def backward(ctx, g):
        row_offsets, cols = ctx.saved_tensors
        R, K = ctx.shape
        return _native.support_scatter(g.contiguous(), row_offsets, cols, R, K), None, None, None


class _GatherRows(torch.autograd.Function):
    """table[rows] for the compaction's (sorted, row-major) rows.  The backward sums each row's
    entries in order with one kernel (smlvm_support_segment_sum) instead of torch's index_select
    backward, which adds with float atomics in an order that changes from run to run."""

    @staticmethod
    def forward(ctx, table, rows, row_offsets):
        ctx.save_for_backward(row_offsets)
        ctx.R = table.shape[0]
        return table.index_select(0, rows)

    @staticmethod
    def backward(ctx, g):
        (row_offsets,) = ctx.saved_tensors
        return _native.support_segment_sum(g.contiguous(), row_offsets, ctx.R), None, None


def gather_rows(table, rows, row_offsets):
    """table [R, ...] gathered at the compaction's rows [N] (int64, row-major order) with a
    deterministic backward; row_offsets [R+1] are the compaction's."""
    return _GatherRows.apply(table, rows, row_offsets)


def gather_support(p, row_offsets, cols, vals):
    """The compacted probabilities `vals` (== p[rows, cols], row-major) as a differentiable function
    of p [R, K]."""
    return _GatherSupport.apply(p, row_offsets, cols, vals)


class _SparseMAP(torch.autograd.Function):
    """Batched bit-vector SparseMAP (sparsemap.py:12-44 for every sample of the batch at once)."""

    @staticmethod
    def forward(ctx, scores, init_scores, budget, max_iter):
        out = _native.sparsemap_fwd(scores.detach(), init_scores, budget, max_iter)
        ctx.n = scores.shape[1]
        ctx.save_for_backward(out["aset_bits"], out["alen"], out["inv_a"])
        for k in ("aset_bits", "alen", "iters", "status"):
            ctx.mark_non_differentiable(out[k])
        return out["p"], out["aset_bits"], out["alen"], out["iters"], out["status"]

    @staticmethod
    def backward(ctx, dp, *_):
        aset_bits, alen, inv_a = ctx.saved_tensors
        return _native.sparsemap_bwd(dp.contiguous(), aset_bits, alen, inv_a, ctx.n), None, None, None


class _LowRankCE(torch.autograd.Function):
    """sum_f cross_entropy(x[n,f,:] @ w.T + b, y[n,f]) without the [N,F,C] logits
    (experiments/bit_vector-vae/archs.py:26-29 + train.py:361-371)."""

    @staticmethod
    def forward(ctx, x, w, b, y):
        # when x wants a gradient the forward pass also leaves d loss / d x (R packed FMAs more per
        # logit): the backward's dX is then an elementwise product, not a third pass of exponentials
        ctx.has_dxdir = bool(ctx.needs_input_grad[0]) and x.numel() > 0
        if ctx.has_dxdir:
            loss, lse, _, dxdir = _native.lowrank_ce_fwd(x, w, b, y, want_dxdir=True)
            ctx.save_for_backward(x, w, b, y, lse, dxdir)
        else:
            loss, lse, _ = _native.lowrank_ce_fwd(x, w, b, y)
            ctx.save_for_backward(x, w, b, y, lse)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        x, w, b, y, lse = ctx.saved_tensors[:5]
        gloss = gloss.contiguous().to(torch.float32)
        want_dw = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        if ctx.has_dxdir:
            dx = ctx.saved_tensors[5] * gloss.view(-1, 1, 1) if ctx.needs_input_grad[0] else None
            dw = db = None
            if want_dw:
                _, dw, db = _native.lowrank_ce_bwd(x, w, b, y, lse, gloss, want_dx=False, want_dw=True)
            return dx, dw, db, None
        dx, dw, db = _native.lowrank_ce_bwd(x, w, b, y, lse, gloss,
                                            want_dx=ctx.needs_input_grad[0], want_dw=want_dw)
        return dx, dw, db, None


def lowrank_categorical_nll(x, w, b, y):
    """x [N,F,R], w [C,R], b [C], y [N,F] integer labels -> per-row loss [N]."""
    return _LowRankCE.apply(x.contiguous(), w.contiguous(), b.contiguous(), y)


_NARROW = (torch.float16, torch.bfloat16)


def sparsemax(X, dim=-1):
    """entmax.sparsemax replacement: Euclidean projection of each slice along `dim` onto the simplex.
    The kernels compute in fp32; half / bfloat16 inputs (autocast) are widened, and the result
    comes back in the input's dtype (float64 is refused rather than silently narrowed)."""
    if X.dtype in _NARROW:
        return sparsemax(X.float(), dim).to(X.dtype)
    if dim < 0:
        dim += X.dim()
    if dim != X.dim() - 1:
        return _Sparsemax.apply(X.transpose(dim, -1).contiguous())[0].transpose(dim, -1)
    p, cols, patterns = _Sparsemax.apply(X.contiguous())
    if p.dim() == 2:
        _tag_support(p, cols, patterns)
    return p


def entmax15(X, dim=-1):
    """entmax.entmax15 replacement: p = max(X/2 - tau, 0)^2 along `dim`, sum p = 1 (fp32 kernels;
    half / bfloat16 inputs are widened as in `sparsemax`)."""
    if X.dtype in _NARROW:
        return entmax15(X.float(), dim).to(X.dtype)
    if dim < 0:
        dim += X.dim()
    if dim != X.dim() - 1:
        return _Entmax15.apply(X.transpose(dim, -1).contiguous()).transpose(dim, -1)
    return _Entmax15.apply(X.contiguous())


SUPPORT_ATTR = "_smlvm_support_cols"


def _tag_support(p, cols, patterns=None):
    """Hang the forward's support lists on the probability tensor itself, so that a later
    `_native.support_compact(p.detach(), support_cols=support_cols_of(p))` does not have to scan p
    again.  The tag lives on this tensor object only (views, detach() and arithmetic drop it) and
    records the tensor's storage and version counter: an in-place change of p afterwards (masking,
    renormalising, clamp_) makes the lists stale, and support_cols_of then returns None.
    patterns: the forward's 16-bit-per-lane non-zero patterns, when it wrote them (rows without a
    list are then counted from those instead of from p: `support_mask_of`)."""
    setattr(p, SUPPORT_ATTR, (cols, p.data_ptr(), p._version, patterns))


def _support_tag(p):
    tag = getattr(p, SUPPORT_ATTR, None)
    if tag is None:
        return None
    cols, ptr, version, patterns = tag
    if p.dim() != 2 or cols.shape[0] != p.shape[0] or p.data_ptr() != ptr or p._version != version:
        return None
    return cols, patterns


def support_cols_of(p):
    tag = _support_tag(p)
    return None if tag is None else tag[0]


def support_mask_of(p):
    """The forward's patterns for `_native.support_compact(..., support_mask=...)`, or None."""
    tag = _support_tag(p)
    return None if tag is None else tag[1]


def sparsemax_with_entropy(X):
    """X [B,K] -> (p [B,K], entropy [B], supp_size [B] int32)."""
    p, ent, supp, cols, patterns = _SparsemaxEntropy.apply(X.contiguous())
    _tag_support(p, cols, patterns)
    return p, ent, supp


def entropy(p):
    return _Entropy.apply(p.contiguous())


def weighted_loss(vals, comps, inv_b):
    """inv_b * sum_j vals[j]*comps[j] as a 0-d tensor (differentiable in both)."""
    return _WeightedLossReduce.apply(vals.contiguous(), comps.contiguous(), float(inv_b))


def sparsemap_batched(scores, init_scores=None, budget=0, max_iter=300):
    """scores [B,n] -> (p [B,n+1] differentiable, aset_bits, alen, iters, status)."""
    return _SparseMAP.apply(scores.contiguous(), init_scores, int(budget), int(max_iter))
