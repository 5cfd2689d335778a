"""ctypes binding of ``libsparsemarg_hip.so`` (C ABI: ``include/sparsemarg.h``) for torch tensors.

PyTorch is only plumbing here: it owns device memory and the current HIP stream; every function
below hands raw device pointers, sizes and that stream to the C ABI and returns torch tensors that
wrap the outputs.  There is NO fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libsparsemarg_hip.so")

_lib = None

ABI_SYMBOLS = [
    "smlvm_abi_version", "smlvm_error_string", "smlvm_last_hip_error", "smlvm_graph_memsets_to_kernels", "smlvm_support_segment_sum", "smlvm_support_compact_fused_masks",
    "smlvm_sparsemax_fwd", "smlvm_sparsemax_bwd", "smlvm_sparsemax_fwd_cols",
    "smlvm_sparsemax_bwd_cols", "smlvm_entropy", "smlvm_entmax15_fwd", "smlvm_entmax15_bwd",
    "smlvm_entmax15_fwd_lists", "smlvm_entmax15_bwd_lists",
    "smlvm_support_count", "smlvm_support_offsets_workspace_bytes", "smlvm_support_offsets",
    "smlvm_support_offsets_noalloc",
    "smlvm_sparsemax_support_mask_bytes", "smlvm_support_compact", "smlvm_support_count_cols", "smlvm_support_compact_cols",
    "smlvm_sparsemax_fwd_lists", "smlvm_support_compact_fused_workspace_bytes", "smlvm_support_compact_fused",
    "smlvm_weighted_loss_reduce_workspace_bytes",
    "smlvm_weighted_loss_reduce_fwd", "smlvm_weighted_loss_reduce_bwd",
    "smlvm_support_entropy_fwd", "smlvm_support_entropy_bwd", "smlvm_support_scatter",
    "smlvm_binary_topk_batched", "smlvm_sparsemap_fwd_workspace_bytes", "smlvm_sparsemap_fwd",
    "smlvm_sparsemap_bwd", "smlvm_sparsemap_seq_fwd", "smlvm_sparsemap_seq_bwd",
    "smlvm_unpack_configs",
    "smlvm_lowrank_ce_fwd", "smlvm_lowrank_ce_fwd_dxdir", "smlvm_lowrank_ce_bwd_workspace_bytes", "smlvm_lowrank_ce_bwd",
    "smlvm_segmented_logsumexp",
]


class NativeError(RuntimeError):
    pass


def lib():
    """Load the HIP library (once). Raises NativeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C sparse-marginalization-lvm_amd/csrc`). There is no "
                "CPU fallback for the hot path.")
        L = ctypes.CDLL(LIB_PATH)
        L.smlvm_error_string.restype = ctypes.c_char_p
        L.smlvm_support_offsets_workspace_bytes.restype = ctypes.c_size_t
        L.smlvm_sparsemax_support_mask_bytes.restype = ctypes.c_size_t
        L.smlvm_sparsemax_support_mask_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
        L.smlvm_support_offsets_workspace_bytes.argtypes = [ctypes.c_int64]
        L.smlvm_support_compact_fused_workspace_bytes.restype = ctypes.c_size_t
        L.smlvm_support_compact_fused_workspace_bytes.argtypes = [ctypes.c_int64]
        L.smlvm_weighted_loss_reduce_workspace_bytes.restype = ctypes.c_size_t
        L.smlvm_weighted_loss_reduce_workspace_bytes.argtypes = [ctypes.c_int64]
        L.smlvm_sparsemap_fwd_workspace_bytes.restype = ctypes.c_size_t
        L.smlvm_sparsemap_fwd_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
        L.smlvm_lowrank_ce_bwd_workspace_bytes.restype = ctypes.c_size_t
        L.smlvm_lowrank_ce_bwd_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32,
                                                           ctypes.c_int32, ctypes.c_int32]
        if L.smlvm_abi_version() != 2:
            raise NativeError("libsparsemarg_hip.so: ABI version mismatch")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        L = lib()
        msg = L.smlvm_error_string(rc).decode()
        raise NativeError(f"{what}: {msg} (code {rc}, hip error {L.smlvm_last_hip_error()})")


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# Host cost of a call: at the reference's batch size a step is a chain of ~10 of these wrappers and
# the GPU waits for each.  The raw stream handle and the device check go straight to torch._C (no
# Stream object, no context manager when the tensor's device is already current): 11.3 -> ~7 us per
# call (tools/t_host_overhead.py; a torch elementwise op costs 4.3 us).
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Noop:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOOP = _Noop()


def _on(dev):
    """Context that makes `dev` the current device -- nothing at all when it already is."""
    if _raw_device is not None and dev.index is not None and _raw_device() == dev.index:
        return _NOOP
    return torch.cuda.device(dev)


def _need_gpu(t, dtype, name):
    if not t.is_cuda:
        raise NativeError(f"{name}: expected a GPU tensor (the hot path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_f32 = ctypes.c_float


# ----------------------------------------------------------------------------------- sparsemax
SUPPORT_COLS_CAP = 32   # slots per row of the forward's support list (128 B per row)
SUPPORT_COLS_CAP_WIDE = 128   # rows wider than a wave's tile (K > 4096: 16 KB and more per row)


def support_cols_cap(K):
    """List capacity the forward is asked for: a list stops paying near 45 entries at K <= 4096
    (measured with the round-1 backward: one 64-byte gather and one scattered store per entry against
    a 16 KB row; rows beyond it carry their pattern instead, which the backward compacts itself), far later for the
    segment backward of wider rows."""
    return SUPPORT_COLS_CAP if K <= 4096 else SUPPORT_COLS_CAP_WIDE


def sparsemax_fwd(x, want_nnz=False, want_tau=False, want_entropy=False, want_cols=False,
                  want_mask=False, want_vals=False, cols_cap=None):
    """x [R,K] f32 -> p [R,K], supp_size [R] i32, and the optional nnz/tau/entropy (or None).
    cols_cap: slots per row of the lists (default support_cols_cap(K); the C ABI states the limits).
    want_cols=True appends the support lists [R, support_cols_cap(K)] i32 for sparsemax_bwd_cols;
    want_mask=True (with want_cols) appends the support-mask buffer as well (None when the shape
    has none: only 1024 < K <= 4096), which lets the backward skip p for rows with more non-zeros
    than a list holds.  want_vals=True (with want_cols, K <= 4096) appends the value lists
    [R, cap] f32 (the probabilities of the listed columns, slot for slot; None for wider rows) that
    let `support_compact` run without gathering from p."""
    x = _need_gpu(x, torch.float32, "x")
    R, K = x.shape
    p = torch.empty_like(x)
    supp = torch.empty(R, dtype=torch.int32, device=x.device)
    nnz = torch.empty(R, dtype=torch.int32, device=x.device) if want_nnz else None
    tau = torch.empty(R, dtype=torch.float32, device=x.device) if want_tau else None
    ent = torch.empty(R, dtype=torch.float32, device=x.device) if want_entropy else None
    if not want_cols:
        with _on(x.device):
            _check(lib().smlvm_sparsemax_fwd(_p(x), _p(p), _p(supp), _p(nnz), _p(tau), _p(ent),
                                             _i64(R), _i32(K), _stream()), "smlvm_sparsemax_fwd")
        return p, supp, nnz, tau, ent
    cap = support_cols_cap(K) if cols_cap is None else int(cols_cap)
    cols = torch.empty((R, cap), dtype=torch.int32, device=x.device)
    mask = None
    if want_mask:
        nbytes = lib().smlvm_sparsemax_support_mask_bytes(_i64(R), _i32(K))
        if nbytes:
            # zeroed: a row is only ever marked -3 after its bits were written, this is defence
            mask = torch.zeros(nbytes // 4, dtype=torch.int32, device=x.device)
    vals = torch.empty((R, cap), dtype=torch.float32, device=x.device) if (want_vals and K <= 4096) else None
    with _on(x.device):
        if vals is not None:
            _check(lib().smlvm_sparsemax_fwd_lists(_p(x), _p(p), _p(supp), _p(nnz), _p(tau), _p(ent),
                                                   _p(cols), _p(vals), _i32(cap), _p(mask), _i64(R),
                                                   _i32(K), _stream()), "smlvm_sparsemax_fwd_lists")
        else:
            _check(lib().smlvm_sparsemax_fwd_cols(_p(x), _p(p), _p(supp), _p(nnz), _p(tau), _p(ent),
                                                  _p(cols), _i32(cap), _p(mask), _i64(R),
                                                  _i32(K), _stream()), "smlvm_sparsemax_fwd_cols")
    out = (p, supp, nnz, tau, ent, cols) + ((mask,) if want_mask else ()) + ((vals,) if want_vals else ())
    return out


def sparsemax_bwd_cols(dy, cols, p, supp, mask=None):
    """Backward from the forward's support lists (p is only read for rows that have neither a list
    nor, with `mask` from the same forward, a support mask)."""
    dy = _need_gpu(dy, torch.float32, "dy")
    p = _need_gpu(p, torch.float32, "p")
    cols = _need_gpu(cols, torch.int32, "cols")
    R, K = p.shape
    if mask is not None:
        mask = _need_gpu(mask, torch.int32, "mask")
        if mask.numel() * 4 < lib().smlvm_sparsemax_support_mask_bytes(_i64(R), _i32(K)):
            raise ValueError("support mask too small for this shape")
    dx = torch.empty_like(p)
    with _on(p.device):
        _check(lib().smlvm_sparsemax_bwd_cols(_p(dy), _p(cols), _i32(cols.shape[1]), _p(mask), _p(p),
                                              _p(supp), _p(dx), _i64(R), _i32(K), _stream()),
               "smlvm_sparsemax_bwd_cols")
    return dx


def sparsemax_bwd(dy, p, supp):
    dy = _need_gpu(dy, torch.float32, "dy")
    p = _need_gpu(p, torch.float32, "p")
    supp = _need_gpu(supp, torch.int32, "supp_size")
    R, K = p.shape
    dx = torch.empty_like(p)
    with _on(p.device):
        _check(lib().smlvm_sparsemax_bwd(_p(dy), _p(p), _p(supp), _p(dx), _i64(R), _i32(K),
                                         _stream()), "smlvm_sparsemax_bwd")
    return dx


def entropy(p):
    p = _need_gpu(p, torch.float32, "p")
    R, K = p.shape
    out = torch.empty(R, dtype=torch.float32, device=p.device)
    with _on(p.device):
        _check(lib().smlvm_entropy(_p(p), _p(out), _i64(R), _i32(K), _stream()), "smlvm_entropy")
    return out


def entmax15_fwd(x, want_nnz=False, want_tau=False):
    """x [R,K] -> (p, nnz | None, tau | None); 1.5-entmax over the last axis."""
    x = _need_gpu(x, torch.float32, "x")
    R, K = x.shape
    p = torch.empty_like(x)
    nnz = torch.empty(R, dtype=torch.int32, device=x.device) if want_nnz else None
    tau = torch.empty(R, dtype=torch.float32, device=x.device) if want_tau else None
    with _on(x.device):
        _check(lib().smlvm_entmax15_fwd(_p(x), _p(p), _p(nnz), _p(tau), _i64(R), _i32(K), _stream()),
               "smlvm_entmax15_fwd")
    return p, nnz, tau


ENTMAX15_LIST_CAP = 64   # slots per row of entmax15's support lists (its supports are ~4x sparsemax's)


def entmax15_has_lists(K):
    """The list-carrying pair exists for rows held by one wave."""
    return 512 < K <= 4096


def entmax15_fwd_lists(x, cap=ENTMAX15_LIST_CAP):
    """x [R,K], 512 < K <= 4096 -> (p, support_cols [R,cap] i32, support_vals [R,cap] f32 = sqrt(p))."""
    x = _need_gpu(x, torch.float32, "x")
    R, K = x.shape
    p = torch.empty_like(x)
    cols = torch.empty((R, cap), dtype=torch.int32, device=x.device)
    vals = torch.empty((R, cap), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _check(lib().smlvm_entmax15_fwd_lists(_p(x), _p(p), None, None, _p(cols), _p(vals), _i32(cap),
                                              _i64(R), _i32(K), _stream()), "smlvm_entmax15_fwd_lists")
    return p, cols, vals


def entmax15_bwd_lists(dy, p, cols, vals):
    dy = _need_gpu(dy, torch.float32, "dy")
    p = _need_gpu(p, torch.float32, "p")
    cols = _need_gpu(cols, torch.int32, "support_cols")
    vals = _need_gpu(vals, torch.float32, "support_vals")
    R, K = p.shape
    if tuple(cols.shape) != tuple(vals.shape) or cols.shape[0] != R:
        raise ValueError("support_cols / support_vals must be [R, cap]")
    dx = torch.empty_like(p)
    with _on(p.device):
        _check(lib().smlvm_entmax15_bwd_lists(_p(dy), _p(p), _p(cols), _p(vals), _i32(cols.shape[1]), _p(dx),
                                              _i64(R), _i32(K), _stream()), "smlvm_entmax15_bwd_lists")
    return dx


def entmax15_bwd(dy, p):
    dy = _need_gpu(dy, torch.float32, "dy")
    p = _need_gpu(p, torch.float32, "p")
    R, K = p.shape
    dx = torch.empty_like(p)
    with _on(p.device):
        _check(lib().smlvm_entmax15_bwd(_p(dy), _p(p), _p(dx), _i64(R), _i32(K), _stream()),
               "smlvm_entmax15_bwd")
    return dx


# ----------------------------------------------------------------------------------- compaction
def support_count(p):
    p = _need_gpu(p, torch.float32, "p")
    R, K = p.shape
    nnz = torch.empty(R, dtype=torch.int32, device=p.device)
    with _on(p.device):
        _check(lib().smlvm_support_count(_p(p), _p(nnz), _i64(R), _i32(K), _stream()),
               "smlvm_support_count")
    return nnz


def support_offsets(nnz):
    nnz = _need_gpu(nnz, torch.int32, "nnz")
    R = nnz.shape[0]
    off = torch.empty(R + 1, dtype=torch.int64, device=nnz.device)
    nbytes = lib().smlvm_support_offsets_workspace_bytes(R)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=nnz.device)
    with _on(nnz.device):
        _check(lib().smlvm_support_offsets(_p(nnz), _p(off), _i64(R), _p(ws),
                                           ctypes.c_size_t(nbytes), _stream()),
               "smlvm_support_offsets")
    return off


def support_row_offsets(p, nnz=None, support_cols=None):
    """row_offsets [R+1] int64 of the compaction of p (count + exclusive scan, no host sync)."""
    p = _need_gpu(p, torch.float32, "p")
    R, K = p.shape
    if support_cols is not None and support_cols.shape[1] > 64:
        support_cols = None
    if nnz is None:
        if support_cols is None:
            nnz = support_count(p)
        else:
            support_cols = _need_gpu(support_cols, torch.int32, "support_cols")
            nnz = torch.empty(R, dtype=torch.int32, device=p.device)
            with _on(p.device):
                _check(lib().smlvm_support_count_cols(_p(p), _p(support_cols), _i32(support_cols.shape[1]),
                                                      _p(nnz), _i64(R), _i32(K), _stream()),
                       "smlvm_support_count_cols")
    return support_offsets(nnz)


FUSED_MAX_BOUND = 1 << 27   # entries: the largest upper bound the single-pass form allocates for
FUSED_SMALL = 1 << 22       # R * K up to which the bound is simply R * K


def support_compact_fused(p, support_cols=None, support_vals=None, want_rows=True, want_cols=True,
                          want_vals=True, n_cap=None, host_offsets=False, static=False, support_mask=None):
    """Count + offsets + compaction in ONE launch (smlvm_support_compact_fused) and one host sync
    (N).  The outputs are sized by an upper bound on N -- R * cap when the rows have lists, R * K
    for small matrices -- and returned as views of length N; should N exceed the bound (rows
    without a list in a list-fed call) the entries are redone from the offsets.  Returns
    (row_offsets, rows, cols, vals) like support_compact, or None when no affordable bound exists.
    host_offsets=True appends the offsets as a host numpy array, read back in the SAME transfer as
    N (the callers that need per-row counts on the host -- SparseMAPWrapper's `support` / `idxs`
    lists -- then pay one device-to-host copy per step instead of two).
    support_mask: the same forward's 16-bit-per-lane patterns (`functional.support_mask_of(p)`: only
    when the forward wrote them): rows without a list are counted from those instead of from p
    (smlvm_support_compact_fused_masks).
    static=True: no host sync at all and shapes that do not depend on the data (what a step captured
    in a HIP graph needs, lvmhelpers/graphs.py): the outputs keep their full length n_cap (default
    R * K, which cannot overflow), the entries behind the N real ones are (row 0, col 0, val 0.0),
    and a fifth result, `n_total` [1] int64 on the device, is the true N.  With a smaller n_cap
    the entries beyond it are dropped and the offsets are clipped to it (everything downstream
    stays in bounds): n_total > n_cap says that this happened -- the caller's to check."""
    p = _need_gpu(p, torch.float32, "p")
    R, K = p.shape
    dev = p.device
    cap = 0
    if support_cols is not None:
        support_cols = _need_gpu(support_cols, torch.int32, "support_cols")
        cap = support_cols.shape[1]
        if support_cols.dim() != 2 or support_cols.shape[0] != R:
            raise ValueError("support_cols must be [R, cap]")
        if cap not in (16, 32, 64):
            support_cols, support_vals, cap = None, None, 0
    if support_vals is not None:
        support_vals = _need_gpu(support_vals, torch.float32, "support_vals")
        if support_cols is None or tuple(support_vals.shape) != tuple(support_cols.shape):
            raise ValueError("support_vals must match support_cols")
    if static:
        n_cap = R * K if n_cap is None else max(1, min(int(n_cap), R * K))
    if n_cap is None:
        # small matrices: R * K can never overflow; with lists: twice the list capacity per row
        # (rows without a list -- more than cap non-zeros -- are paid for by the rows with short
        # ones; should N still exceed it, the entries are redone below)
        if R * K <= FUSED_SMALL or support_cols is None:
            n_cap = R * K
        else:
            n_cap = R * min(K, 2 * cap)
        if n_cap > FUSED_MAX_BOUND:
            return None
    nbytes = lib().smlvm_support_compact_fused_workspace_bytes(_i64(R))
    both = torch.empty(nbytes // 8 + R + 1, dtype=torch.int64, device=dev)   # workspace | offsets
    ws, off = both[:nbytes // 8], both[nbytes // 8:]
    alloc = torch.zeros if static else torch.empty
    rows = alloc(n_cap, dtype=torch.int32, device=dev) if want_rows else None
    cols = alloc(n_cap, dtype=torch.int32, device=dev) if want_cols else None
    vals = alloc(n_cap, dtype=torch.float32, device=dev) if want_vals else None
    use_masks = (support_mask is not None and support_cols is not None and 32 < K <= 1024 and K % 4 == 0)
    if use_masks:
        support_mask = _need_gpu(support_mask, torch.int32, "support_mask")
        if support_mask.numel() * 4 < lib().smlvm_sparsemax_support_mask_bytes(_i64(R), _i32(K)):
            raise ValueError("support mask too small for this shape")
    with _on(dev):
        if use_masks:
            _check(lib().smlvm_support_compact_fused_masks(
                _p(p), _p(support_cols), _p(support_vals), _i32(cap), _p(support_mask), _p(off), _p(rows),
                _p(cols), _p(vals), _i64(n_cap), _i64(R), _i32(K), _p(ws), ctypes.c_size_t(nbytes),
                _stream()), "smlvm_support_compact_fused_masks")
        else:
            _check(lib().smlvm_support_compact_fused(_p(p), _p(support_cols), _p(support_vals), _i32(cap),
                                                     _p(off), _p(rows), _p(cols), _p(vals), _i64(n_cap),
                                                     _i64(R), _i32(K), _p(ws), ctypes.c_size_t(nbytes),
                                                     _stream()), "smlvm_support_compact_fused")
    if static:
        n_total = ws[2:3].clone()
        if n_cap < R * K:
            off.clamp_(max=n_cap)
        return off, rows, cols, vals, n_total
    if R == 0:
        return (off, rows, cols, vals) + ((off.cpu().numpy(),) if host_offsets else ())
    off_h = None
    if host_offsets:
        both_h = both.cpu().numpy()                        # the one host sync: flags, N and the offsets
        err, N, off_h = int(both_h[1]), int(both_h[2]), both_h[nbytes // 8:]
    else:
        err, N = (int(v) for v in ws[1:3].tolist())        # the one host sync
    if err:
        raise NativeError("smlvm_support_compact_fused: a tile gave up waiting for the tiles before it")
    extra = (off_h,) if host_offsets else ()
    if N > n_cap:   # rows without a list overflowed the bound: the entries again, from the offsets
        return support_compact(p, want_rows=want_rows, want_cols=want_cols, want_vals=want_vals,
                               support_cols=support_cols, row_offsets=off, total=N, fused=False) + extra
    cut = lambda t: None if t is None else t[:N]
    return (off, cut(rows), cut(cols), cut(vals)) + extra


def support_compact(p, nnz=None, want_rows=True, want_cols=True, want_vals=True, support_cols=None,
                    row_offsets=None, total=None, support_vals=None, fused=None, support_mask=None):
    """Row-major (rows, cols, vals) lists of the positive entries of p [R,K] plus row_offsets.

    support_cols: the forward's support lists for this very p (`sparsemax_fwd(..., want_cols=True)`,
    or `functional.support_cols_of(p)`): rows that have a list are counted and compacted from it
    instead of from their K probabilities; the output is the same.  support_vals: the forward's
    value lists (`want_vals=True`): then nothing is gathered from p either.
    fused=False forces the separate count / offsets / compact launches.
    One host sync (reading N = row_offsets[R]) to size the outputs -- none when the caller has the
    offsets already (`row_offsets` from support_row_offsets and `total` = N read on its side).
    """
    p = _need_gpu(p, torch.float32, "p")
    R, K = p.shape
    if row_offsets is None and nnz is None and fused is not False:
        # the single-pass form (one launch, one host sync) whenever an upper bound on N is affordable
        out = support_compact_fused(p, support_cols, support_vals, want_rows, want_cols, want_vals,
                                    support_mask=support_mask)
        if out is not None:
            return out
    cap = 0
    if support_cols is not None:
        support_cols = _need_gpu(support_cols, torch.int32, "support_cols")
        if support_cols.dim() != 2 or support_cols.shape[0] != R:
            raise ValueError("support_cols must be [R, cap]")
        cap = support_cols.shape[1]
        if cap > 64:   # the list-fed compaction kernels take caps up to 64: scan p instead
            support_cols, cap = None, 0
    if row_offsets is not None:
        off = _need_gpu(row_offsets, torch.int64, "row_offsets")
        N = int(total) if total is not None else int(off[R].item())
    else:
        off = support_row_offsets(p, nnz, support_cols)
        N = int(off[R].item())
    dev = p.device
    rows = torch.empty(N, dtype=torch.int32, device=dev) if want_rows else None
    cols = torch.empty(N, dtype=torch.int32, device=dev) if want_cols else None
    vals = torch.empty(N, dtype=torch.float32, device=dev) if want_vals else None
    with _on(dev):
        if support_cols is None:
            _check(lib().smlvm_support_compact(_p(p), _p(off), _p(rows), _p(cols), _p(vals), _i64(R),
                                               _i32(K), _stream()), "smlvm_support_compact")
        else:
            _check(lib().smlvm_support_compact_cols(_p(p), _p(support_cols), _i32(cap), _p(off),
                                                    _p(rows), _p(cols), _p(vals), _i64(R), _i32(K),
                                                    _stream()), "smlvm_support_compact_cols")
    return off, rows, cols, vals


def weighted_loss_reduce_fwd(vals, comps, row_offsets=None, inv_b=1.0, want_rows=False):
    vals = _need_gpu(vals, torch.float32, "vals")
    comps = _need_gpu(comps, torch.float32, "comps")
    N = vals.shape[0]
    if comps.shape[0] != N:
        raise ValueError("vals and comps must have the same length")
    dev = vals.device
    R = 0 if row_offsets is None else row_offsets.shape[0] - 1
    row_out = torch.empty(R, dtype=torch.float32, device=dev) if (want_rows and R > 0) else None
    total = torch.empty(1, dtype=torch.float32, device=dev)
    nbytes = lib().smlvm_weighted_loss_reduce_workspace_bytes(N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with _on(dev):
        _check(lib().smlvm_weighted_loss_reduce_fwd(
            _p(vals), _p(comps), _p(row_offsets), _p(row_out), _p(total), _f32(inv_b), _i64(R),
            _i64(N), _p(ws), ctypes.c_size_t(nbytes), _stream()), "smlvm_weighted_loss_reduce_fwd")
    return total, row_out


def weighted_loss_reduce_bwd(vals, comps, gtotal, inv_b, want_dvals=True, want_dcomps=True):
    vals = _need_gpu(vals, torch.float32, "vals")
    comps = _need_gpu(comps, torch.float32, "comps")
    gtotal = _need_gpu(gtotal.reshape(1), torch.float32, "gtotal")
    N = vals.shape[0]
    dvals = torch.empty_like(vals) if want_dvals else None
    dcomps = torch.empty_like(comps) if want_dcomps else None
    with _on(vals.device):
        _check(lib().smlvm_weighted_loss_reduce_bwd(_p(vals), _p(comps), _p(gtotal), _f32(inv_b),
                                                    _p(dvals), _p(dcomps), _i64(N), _stream()),
               "smlvm_weighted_loss_reduce_bwd")
    return dvals, dcomps


def support_entropy_fwd(vals, inv_b=1.0):
    """-inv_b * sum_{v > 0} v log v over all entries of vals (any shape) -> [1] f32."""
    vals = _need_gpu(vals, torch.float32, "vals")
    N = vals.numel()
    total = torch.empty(1, dtype=torch.float32, device=vals.device)
    nbytes = lib().smlvm_weighted_loss_reduce_workspace_bytes(N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=vals.device)
    with _on(vals.device):
        _check(lib().smlvm_support_entropy_fwd(_p(vals), _i64(N), _f32(inv_b), _p(total), _p(ws),
                                               ctypes.c_size_t(nbytes), _stream()), "smlvm_support_entropy_fwd")
    return total


def support_entropy_bwd(vals, gtotal, inv_b=1.0):
    vals = _need_gpu(vals, torch.float32, "vals")
    gtotal = _need_gpu(gtotal.reshape(1), torch.float32, "gtotal")
    dvals = torch.empty_like(vals)
    with _on(vals.device):
        _check(lib().smlvm_support_entropy_bwd(_p(vals), _p(gtotal), _f32(inv_b), _p(dvals),
                                               _i64(vals.numel()), _stream()), "smlvm_support_entropy_bwd")
    return dvals


def support_segment_sum(g, row_offsets, R):
    """g [N, ...] -> [R, ...]: per-row sums over each row's compacted entries (row_offsets [R+1]),
    in entry order; entries at or beyond row_offsets[R] (static-shape padding) are ignored."""
    g = _need_gpu(g, torch.float32, "g")
    row_offsets = _need_gpu(row_offsets, torch.int64, "row_offsets")
    if row_offsets.shape[0] != R + 1:
        raise ValueError("support_segment_sum: row_offsets must be [R+1]")
    inner = tuple(g.shape[1:])
    D = 1
    for v in inner:
        D *= v
    out = torch.empty((R,) + inner, dtype=torch.float32, device=g.device)
    if g.shape[0] == 0 or D == 0:
        return out.zero_()
    with _on(g.device):
        _check(lib().smlvm_support_segment_sum(_p(g), _p(row_offsets), _p(out), _i64(R), _i64(D), _stream()),
               "smlvm_support_segment_sum")
    return out


def support_scatter(g, row_offsets, cols, R, K):
    """Dense [R, K] with g[j] at (row of j, cols[j]) and zeros elsewhere (j in row-major compaction
    order, rows given by row_offsets [R+1])."""
    g = _need_gpu(g, torch.float32, "g")
    row_offsets = _need_gpu(row_offsets, torch.int64, "row_offsets")
    cols = _need_gpu(cols, torch.int32, "cols")
    if row_offsets.shape[0] != R + 1 or cols.shape[0] != g.shape[0]:
        raise ValueError("support_scatter: row_offsets must be [R+1], cols and g of one length")
    out = torch.empty((R, K), dtype=torch.float32, device=g.device)
    if g.shape[0] == 0:
        return out.zero_()
    with _on(g.device):
        _check(lib().smlvm_support_scatter(_p(g), _p(row_offsets), _p(cols), _p(out), _i64(R), _i32(K),
                                           _stream()), "smlvm_support_scatter")
    return out


# ----------------------------------------------------------------------------------- top-k
def binary_topk_batched(scores, k, want_scores=False):
    scores = _need_gpu(scores, torch.float32, "scores")
    B, n = scores.shape
    cfg = torch.empty((B, k, n), dtype=torch.float32, device=scores.device)
    top = torch.empty((B, k), dtype=torch.float32, device=scores.device) if want_scores else None
    with _on(scores.device):
        _check(lib().smlvm_binary_topk_batched(_p(scores), _p(cfg), _p(top), _i64(B), _i32(n),
                                               _i32(k), _stream()), "smlvm_binary_topk_batched")
    return cfg, top


# ----------------------------------------------------------------------------------- SparseMAP
def sparsemap_fwd(scores, init_scores=None, budget=0, max_iter=300):
    """scores [B,n] f32 (GPU), init_scores [B,2n] f64 (GPU) or None.

    Returns dict(aset_bits [B,cap,W] i32 (bit pattern of uint32), p [B,cap] f32, alen [B] i32,
    inv_a [B,ld,ld] f64, iters [B] i32, status [B] i32).
    """
    scores = _need_gpu(scores, torch.float32, "scores")
    B, n = scores.shape
    cap, W, ld = n + 1, (n + 31) // 32, n + 2
    dev = scores.device
    if init_scores is not None:
        init_scores = _need_gpu(init_scores, torch.float64, "init_scores")
        if tuple(init_scores.shape) != (B, 2 * n):
            raise ValueError("init_scores must be [B, 2n]")
    out = dict(
        aset_bits=torch.empty((B, cap, W), dtype=torch.int32, device=dev),
        p=torch.empty((B, cap), dtype=torch.float32, device=dev),
        alen=torch.empty(B, dtype=torch.int32, device=dev),
        inv_a=torch.empty((B, ld, ld), dtype=torch.float64, device=dev),
        iters=torch.empty(B, dtype=torch.int32, device=dev),
        status=torch.empty(B, dtype=torch.int32, device=dev))
    ws_bytes = lib().smlvm_sparsemap_fwd_workspace_bytes(B, n, 1)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    with _on(dev):
        _check(lib().smlvm_sparsemap_fwd(
            _p(scores), _p(init_scores), _i32(int(budget)), _i32(int(max_iter)), _i64(B), _i32(n),
            _p(out["aset_bits"]), _p(out["p"]), _p(out["alen"]), _p(out["inv_a"]),
            _p(out["iters"]), _p(out["status"]), _p(ws), ctypes.c_size_t(ws_bytes), _stream()),
            "smlvm_sparsemap_fwd")
    return out


def sparsemap_seq_fwd(scores, trans, init_scores=None, max_iter=300):
    """Binary sequence factor: scores [B,n] f32, trans [B,n-1] f32 edge potentials of the (1 -> 1)
    transitions, init_scores [B,2n] f64 or None.  Same outputs as sparsemap_fwd."""
    scores = _need_gpu(scores, torch.float32, "scores")
    trans = _need_gpu(trans, torch.float32, "trans")
    B, n = scores.shape
    if tuple(trans.shape) != (B, n - 1):
        raise ValueError("trans must be [B, n-1]")
    cap, W, ld = n + 1, (n + 31) // 32, n + 2
    dev = scores.device
    if init_scores is not None:
        init_scores = _need_gpu(init_scores, torch.float64, "init_scores")
        if tuple(init_scores.shape) != (B, 2 * n):
            raise ValueError("init_scores must be [B, 2n]")
    out = dict(
        aset_bits=torch.empty((B, cap, W), dtype=torch.int32, device=dev),
        p=torch.empty((B, cap), dtype=torch.float32, device=dev),
        alen=torch.empty(B, dtype=torch.int32, device=dev),
        inv_a=torch.empty((B, ld, ld), dtype=torch.float64, device=dev),
        iters=torch.empty(B, dtype=torch.int32, device=dev),
        status=torch.empty(B, dtype=torch.int32, device=dev))
    ws_bytes = lib().smlvm_sparsemap_fwd_workspace_bytes(B, n, 1)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    with _on(dev):
        _check(lib().smlvm_sparsemap_seq_fwd(
            _p(scores), _p(trans), _p(init_scores), _i32(int(max_iter)), _i64(B), _i32(n),
            _p(out["aset_bits"]), _p(out["p"]), _p(out["alen"]), _p(out["inv_a"]),
            _p(out["iters"]), _p(out["status"]), _p(ws), ctypes.c_size_t(ws_bytes), _stream()),
            "smlvm_sparsemap_seq_fwd")
    return out


def sparsemap_seq_bwd(dp, aset_bits, alen, inv_a, n):
    """dp [B,n+1] f32 -> (dscores [B,n], dtrans [B,n-1]) f32."""
    dp = _need_gpu(dp, torch.float32, "dp")
    B = dp.shape[0]
    dscores = torch.empty((B, n), dtype=torch.float32, device=dp.device)
    dtrans = torch.empty((B, n - 1), dtype=torch.float32, device=dp.device)
    with _on(dp.device):
        _check(lib().smlvm_sparsemap_seq_bwd(_p(dp), _p(aset_bits), _p(alen), _p(inv_a), _i64(B),
                                             _i32(n), _p(dscores), _p(dtrans), _stream()),
               "smlvm_sparsemap_seq_bwd")
    return dscores, dtrans


def sparsemap_bwd(dp, aset_bits, alen, inv_a, n):
    dp = _need_gpu(dp, torch.float32, "dp")
    B = dp.shape[0]
    dscores = torch.empty((B, n), dtype=torch.float32, device=dp.device)
    with _on(dp.device):
        _check(lib().smlvm_sparsemap_bwd(_p(dp), _p(aset_bits), _p(alen), _p(inv_a), _i64(B),
                                         _i32(n), _p(dscores), _stream()), "smlvm_sparsemap_bwd")
    return dscores


def unpack_configs(aset_bits, rows, cols, n):
    N = rows.shape[0]
    out = torch.empty((N, n), dtype=torch.float32, device=aset_bits.device)
    with _on(aset_bits.device):
        _check(lib().smlvm_unpack_configs(_p(aset_bits), _p(rows), _p(cols), _p(out), _i64(N),
                                          _i32(n), _stream()), "smlvm_unpack_configs")
    return out


_Y_KINDS = {torch.uint8: 0, torch.int32: 1, torch.int64: 2}


def _labels(y, N, F):
    if not y.is_cuda:
        raise NativeError("labels: expected a GPU tensor (the hot path has no CPU fallback)")
    if y.dtype not in _Y_KINDS:
        y = y.to(torch.int64)
    y = y.reshape(N, F).contiguous()
    return y, _Y_KINDS[y.dtype]


def lowrank_ce_fwd(x, w, b, y, want_dxdir=False):
    """x [N,F,R] f32, w [C,R] f32, b [C] f32, y [N,F] labels -> (loss [N], lse [N,F], nll [N,F]);
    want_dxdir=True appends dxdir [N,F,R] = d loss[n] / d x[n,f,:] (smlvm_lowrank_ce_fwd_dxdir)."""
    x = _need_gpu(x, torch.float32, "x")
    w = _need_gpu(w, torch.float32, "w")
    b = _need_gpu(b, torch.float32, "b")
    N, F, R = x.shape
    C = w.shape[0]
    if tuple(w.shape) != (C, R) or tuple(b.shape) != (C,):
        raise ValueError("w must be [C,R] and b [C]")
    y, kind = _labels(y, N, F)
    dev = x.device
    lse = torch.empty((N, F), dtype=torch.float32, device=dev)
    nll = torch.empty((N, F), dtype=torch.float32, device=dev)
    loss = torch.empty(N, dtype=torch.float32, device=dev)
    if want_dxdir:
        dxdir = torch.empty_like(x)
        with _on(dev):
            _check(lib().smlvm_lowrank_ce_fwd_dxdir(_p(x), _p(w), _p(b), _p(y), _i32(kind), _p(lse), _p(nll),
                                                    _p(loss), _p(dxdir), _i64(N), _i32(F), _i32(C), _i32(R),
                                                    _stream()), "smlvm_lowrank_ce_fwd_dxdir")
        return loss, lse, nll, dxdir
    with _on(dev):
        _check(lib().smlvm_lowrank_ce_fwd(_p(x), _p(w), _p(b), _p(y), _i32(kind), _p(lse), _p(nll),
                                          _p(loss), _i64(N), _i32(F), _i32(C), _i32(R), _stream()),
               "smlvm_lowrank_ce_fwd")
    return loss, lse, nll


def lowrank_ce_bwd(x, w, b, y, lse, gloss, want_dx=True, want_dw=True):
    x = _need_gpu(x, torch.float32, "x")
    w = _need_gpu(w, torch.float32, "w")
    b = _need_gpu(b, torch.float32, "b")
    lse = _need_gpu(lse, torch.float32, "lse")
    gloss = _need_gpu(gloss, torch.float32, "gloss")
    N, F, R = x.shape
    C = w.shape[0]
    y, kind = _labels(y, N, F)
    dev = x.device
    dx = torch.empty_like(x) if want_dx else None
    # (the reduction kernel writes every entry; only an empty batch leaves them untouched)
    dw = (torch.empty_like(w) if N > 0 else torch.zeros_like(w)) if want_dw else None
    db = (torch.empty_like(b) if N > 0 else torch.zeros_like(b)) if want_dw else None
    nbytes = lib().smlvm_lowrank_ce_bwd_workspace_bytes(N, F, C, R) if want_dw else 0
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
    with _on(dev):
        _check(lib().smlvm_lowrank_ce_bwd(_p(x), _p(w), _p(b), _p(y), _i32(kind), _p(lse), _p(gloss),
                                          _p(dx), _p(dw), _p(db), _i64(N), _i32(F), _i32(C), _i32(R),
                                          _p(ws), ctypes.c_size_t(nbytes), _stream()),
               "smlvm_lowrank_ce_bwd")
    return dx, dw, db


def segmented_logsumexp(v, row_offsets, shift=0.0):
    """out[r] = logsumexp(v[row_offsets[r]:row_offsets[r+1]] + shift); -inf for empty segments."""
    v = _need_gpu(v, torch.float32, "v")
    row_offsets = _need_gpu(row_offsets, torch.int64, "row_offsets")
    R = row_offsets.shape[0] - 1
    out = torch.empty(R, dtype=torch.float32, device=v.device)
    with _on(v.device):
        _check(lib().smlvm_segmented_logsumexp(_p(v), _p(row_offsets), _f32(float(shift)), _p(out),
                                               _i64(R), _stream()), "smlvm_segmented_logsumexp")
    return out


# ----------------------------------------------------------------------------------- HIP graphs
def graph_memsets_to_kernels(raw_graph):
    """raw_graph: the hipGraph_t of a capture (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph())
    -> (memset nodes found, replaced by kernel nodes); see include/sparsemarg.h."""
    seen, done = ctypes.c_int32(0), ctypes.c_int32(0)
    _check(lib().smlvm_graph_memsets_to_kernels(ctypes.c_void_p(int(raw_graph)), ctypes.byref(seen),
                                                ctypes.byref(done)), "smlvm_graph_memsets_to_kernels")
    return seen.value, done.value
