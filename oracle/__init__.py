"""ORACLE (test infrastructure, NOT product code).

ctypes bindings for the CPU restatements in ``oracle/*.c`` (``liboracle.so``) and, when it has
been built, for the compiled reference ``oracle/_ref/libref_binary_topk.so`` (the reference's own
``lvmhelpers/binary_topk.h``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product path (``lvmhelpers``) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u32p = ctypes.POINTER(ctypes.c_uint32)


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in os.listdir(_HERE) if f.endswith(".c")):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_support_compact.restype = ctypes.c_int64
        _LIB.oracle_weighted_loss_reduce.restype = ctypes.c_double
    return _LIB


def ref_lib():
    """The compiled reference binary_topk (None when oracle/_ref was not built)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libref_binary_topk.so")
        if not os.path.exists(so):
            return None
        _REF = ctypes.CDLL(so)
    return _REF


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------- sparsemax
def sparsemax_fwd(x):
    """x [R,K] f32 -> (p [R,K] f32, supp_size [R] i32, tau [R] f32, nnz [R] i32)."""
    x = _c32(x)
    R, K = x.shape
    p = np.empty_like(x)
    supp = np.empty(R, np.int32)
    tau = np.empty(R, np.float32)
    nnz = np.empty(R, np.int32)
    rc = lib().oracle_sparsemax_fwd(_ptr(x, _f32p), _ptr(p, _f32p), _ptr(supp, _i32p),
                                    _ptr(tau, _f32p), _ptr(nnz, _i32p),
                                    ctypes.c_int64(R), ctypes.c_int32(K))
    assert rc == 0
    return p, supp, tau, nnz


def sparsemax_bwd(dy, p, supp):
    dy = _c32(dy)
    p = _c32(p)
    supp = np.ascontiguousarray(supp, np.int32)
    R, K = dy.shape
    dx = np.empty_like(dy)
    rc = lib().oracle_sparsemax_bwd(_ptr(dy, _f32p), _ptr(p, _f32p), _ptr(supp, _i32p),
                                    _ptr(dx, _f32p), ctypes.c_int64(R), ctypes.c_int32(K))
    assert rc == 0
    return dx


def entropy(p):
    p = _c32(p)
    R, K = p.shape
    out = np.empty(R, np.float32)
    lib().oracle_entropy(_ptr(p, _f32p), _ptr(out, _f32p), ctypes.c_int64(R), ctypes.c_int32(K))
    return out


def sparsemax_loss(x, target):
    """(loss [R] f32, grad [R,K] f32) of the Fenchel-Young sparsemax loss, reduction='none'."""
    x = _c32(x)
    t = np.ascontiguousarray(target, np.int64)
    R, K = x.shape
    loss = np.empty(R, np.float32)
    grad = np.empty_like(x)
    lib().oracle_sparsemax_loss(_ptr(x, _f32p), _ptr(t, _i64p), _ptr(loss, _f32p),
                                _ptr(grad, _f32p), ctypes.c_int64(R), ctypes.c_int32(K))
    return loss, grad


# ----------------------------------------------------------------------------- entmax15
def entmax15_fwd(x):
    """x [R,K] f32 -> (p [R,K] f32, nnz [R] i32, tau [R] f32 in the frame z = x/2 - max(x/2))."""
    x = _c32(x)
    R, K = x.shape
    p = np.empty_like(x)
    supp = np.empty(R, np.int32)
    tau = np.empty(R, np.float32)
    rc = lib().oracle_entmax15_fwd(_ptr(x, _f32p), _ptr(p, _f32p), _ptr(supp, _i32p), _ptr(tau, _f32p),
                                   ctypes.c_int64(R), ctypes.c_int32(K))
    assert rc == 0
    return p, supp, tau


def entmax15_bwd(dy, p):
    dy, p = _c32(dy), _c32(p)
    R, K = dy.shape
    dx = np.empty_like(dy)
    lib().oracle_entmax15_bwd(_ptr(dy, _f32p), _ptr(p, _f32p), _ptr(dx, _f32p), ctypes.c_int64(R),
                              ctypes.c_int32(K))
    return dx


def entmax15_loss(x, target):
    """(loss [R] f32, grad [R,K] f32) of the Fenchel-Young 1.5-entmax loss, reduction='none'."""
    x = _c32(x)
    t = np.ascontiguousarray(target, np.int64)
    R, K = x.shape
    loss = np.empty(R, np.float32)
    grad = np.empty_like(x)
    rc = lib().oracle_entmax15_loss(_ptr(x, _f32p), _ptr(t, _i64p), _ptr(loss, _f32p), _ptr(grad, _f32p),
                                    ctypes.c_int64(R), ctypes.c_int32(K))
    assert rc == 0
    return loss, grad


# ----------------------------------------------------------------------------- compaction
def support_compact(p):
    """-> (row_offsets [R+1] i64, rows [N] i32, cols [N] i32, vals [N] f32), row-major."""
    p = _c32(p)
    R, K = p.shape
    off = np.empty(R + 1, np.int64)
    N = lib().oracle_support_compact(_ptr(p, _f32p), _ptr(off, _i64p), None, None, None,
                                     ctypes.c_int64(R), ctypes.c_int32(K))
    rows = np.empty(N, np.int32)
    cols = np.empty(N, np.int32)
    vals = np.empty(N, np.float32)
    lib().oracle_support_compact(_ptr(p, _f32p), _ptr(off, _i64p), _ptr(rows, _i32p),
                                 _ptr(cols, _i32p), _ptr(vals, _f32p),
                                 ctypes.c_int64(R), ctypes.c_int32(K))
    return off, rows, cols, vals


def weighted_loss_reduce(vals, comps, row_offsets, inv_b):
    vals = _c32(vals)
    comps = _c32(comps)
    off = np.ascontiguousarray(row_offsets, np.int64)
    R = off.shape[0] - 1
    out = np.empty(R, np.float32)
    tot = lib().oracle_weighted_loss_reduce(_ptr(vals, _f32p), _ptr(comps, _f32p), _ptr(off, _i64p),
                                            _ptr(out, _f32p), ctypes.c_int64(R),
                                            ctypes.c_double(inv_b))
    return out, tot


# ----------------------------------------------------------------------------- binary top-k
def _topk_call(fn, scores, k):
    scores = _c32(scores)
    B, n = scores.shape
    cfg = np.zeros((B, k, n), np.float32)
    top = np.zeros((B, k), np.float32)
    rc = fn(_ptr(scores, _f32p), _ptr(cfg, _f32p), _ptr(top, _f32p),
            ctypes.c_int64(B), ctypes.c_int32(n), ctypes.c_int32(k))
    if rc != 0:
        raise ValueError(f"binary_topk oracle: error {rc} (n={n}, k={k})")
    return cfg, top


def binary_topk_batched(scores, k):
    """Our restatement: scores [B,n] f32 -> (configs [B,k,n] f32 0/1, topscores [B,k] f32)."""
    return _topk_call(lib().oracle_binary_topk_batched, scores, k)


def ref_binary_topk_batched(scores, k):
    """The compiled reference (oracle/_ref). Caller must keep 2 <= k <= 2**n (UB otherwise)."""
    r = ref_lib()
    if r is None:
        raise RuntimeError("oracle/_ref/libref_binary_topk.so not built (needs /root/reference)")
    return _topk_call(r.ref_binary_topk_batched, scores, k)


# ----------------------------------------------------------------------------- SparseMAP
def sparsemap_fwd(scores, init_scores=None, budget=0, max_iter=300):
    """scores [B,n] f32, init_scores [B,2n] f64 or None.

    Returns dict(aset_bits [B,cap,W] u32, p [B,cap] f32, p64 [B,cap] f64, alen [B] i32,
    invA [B,ld,ld] f64, iters [B] i32, status [B] i32) with cap=n+1, W=ceil(n/32), ld=cap+1.
    """
    scores = _c32(scores)
    B, n = scores.shape
    cap, W, ld = n + 1, (n + 31) // 32, n + 2
    init = None if init_scores is None else np.ascontiguousarray(init_scores, np.float64)
    if init is not None:
        assert init.shape == (B, 2 * n)
    out = dict(
        aset_bits=np.zeros((B, cap, W), np.uint32), p=np.zeros((B, cap), np.float32),
        p64=np.zeros((B, cap), np.float64), alen=np.zeros(B, np.int32),
        invA=np.zeros((B, ld, ld), np.float64), iters=np.zeros(B, np.int32),
        status=np.zeros(B, np.int32))
    rc = lib().oracle_sparsemap_fwd(
        _ptr(scores, _f32p), _ptr(init, _f64p), ctypes.c_int32(budget), ctypes.c_int32(max_iter),
        ctypes.c_int64(B), ctypes.c_int32(n), _ptr(out["aset_bits"], _u32p), _ptr(out["p"], _f32p),
        _ptr(out["p64"], _f64p), _ptr(out["alen"], _i32p), _ptr(out["invA"], _f64p),
        _ptr(out["iters"], _i32p), _ptr(out["status"], _i32p))
    if rc != 0:
        raise ValueError(f"sparsemap oracle: error {rc}")
    return out


def sparsemap_bwd(dp, aset_bits, alen, invA, n):
    dp = _c32(dp)
    B = dp.shape[0]
    out = np.zeros((B, n), np.float32)
    rc = lib().oracle_sparsemap_bwd(
        _ptr(dp, _f32p), _ptr(np.ascontiguousarray(aset_bits, np.uint32), _u32p),
        _ptr(np.ascontiguousarray(alen, np.int32), _i32p),
        _ptr(np.ascontiguousarray(invA, np.float64), _f64p),
        ctypes.c_int64(B), ctypes.c_int32(n), _ptr(out, _f32p))
    assert rc == 0
    return out


def sparsemap_seq_fwd(scores, trans, init_scores=None, max_iter=300):
    """Binary sequence factor: scores [B,n] f32, trans [B,n-1] f32 (edge potentials of the 1->1
    transitions), init_scores [B,2n] f64 or None.  Same outputs as sparsemap_fwd."""
    scores, trans = _c32(scores), _c32(trans)
    B, n = scores.shape
    assert trans.shape == (B, n - 1)
    cap, W, ld = n + 1, (n + 31) // 32, n + 2
    init = None if init_scores is None else np.ascontiguousarray(init_scores, np.float64)
    out = dict(
        aset_bits=np.zeros((B, cap, W), np.uint32), p=np.zeros((B, cap), np.float32),
        p64=np.zeros((B, cap), np.float64), alen=np.zeros(B, np.int32),
        invA=np.zeros((B, ld, ld), np.float64), iters=np.zeros(B, np.int32),
        status=np.zeros(B, np.int32))
    rc = lib().oracle_sparsemap_seq_fwd(
        _ptr(scores, _f32p), _ptr(trans, _f32p), _ptr(init, _f64p), ctypes.c_int32(max_iter),
        ctypes.c_int64(B), ctypes.c_int32(n), _ptr(out["aset_bits"], _u32p), _ptr(out["p"], _f32p),
        _ptr(out["p64"], _f64p), _ptr(out["alen"], _i32p), _ptr(out["invA"], _f64p),
        _ptr(out["iters"], _i32p), _ptr(out["status"], _i32p))
    if rc != 0:
        raise ValueError(f"sparsemap sequence oracle: error {rc}")
    return out


def sparsemap_seq_bwd(dp, aset_bits, alen, invA, n):
    """-> (dscores [B,n] f32, dtrans [B,n-1] f32)."""
    dp = _c32(dp)
    B = dp.shape[0]
    ds, dt = np.zeros((B, n), np.float32), np.zeros((B, n - 1), np.float32)
    rc = lib().oracle_sparsemap_seq_bwd(
        _ptr(dp, _f32p), _ptr(np.ascontiguousarray(aset_bits, np.uint32), _u32p),
        _ptr(np.ascontiguousarray(alen, np.int32), _i32p),
        _ptr(np.ascontiguousarray(invA, np.float64), _f64p),
        ctypes.c_int64(B), ctypes.c_int32(n), _ptr(ds, _f32p), _ptr(dt, _f32p))
    assert rc == 0
    return ds, dt


def unpack_bits(aset_bits, n):
    """[..., W] u32 -> [..., n] float32 0/1."""
    a = np.asarray(aset_bits, np.uint32)
    k = np.arange(n)
    return ((a[..., k >> 5] >> (k & 31).astype(np.uint32)) & 1).astype(np.float32)


def torch_sparsemax_fwd_bwd(x, dy):
    """entmax's published sort/cumsum sparsemax and its backward written with torch CPU ops, i.e.
    what the reference's `--mode marg --normalizer sparsemax` runs on a CPU (marg.py:45,51 ->
    entmax.sparsemax).  Multi-threaded by torch; used as the all-cores CPU baseline of bench.py and
    checked bit-exact against the C restatement in tests/test_oracle_sparsemax.py."""
    import torch
    X = torch.from_numpy(x)
    X = X - X.max(dim=-1, keepdim=True)[0]
    srt, _ = torch.sort(X, dim=-1, descending=True)
    cs = srt.cumsum(dim=-1) - 1
    rho = torch.arange(1, X.shape[-1] + 1, dtype=X.dtype)
    supp = (rho * srt > cs).sum(dim=-1, keepdim=True)
    tau = cs.gather(-1, supp - 1) / supp.to(X.dtype)
    p = torch.clamp(X - tau, min=0)
    g = torch.from_numpy(dy).clone()
    g[p == 0] = 0
    v = g.sum(dim=-1, keepdim=True) / supp.to(X.dtype)
    dx = torch.where(p != 0, g - v, g)
    return p.numpy(), supp.squeeze(-1).to(torch.int32).numpy(), dx.numpy()
