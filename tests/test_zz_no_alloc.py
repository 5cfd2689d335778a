"""GPU: the allocation-free entry point smlvm_support_offsets_noalloc (block sums in a static device
array of the library) against smlvm_support_offsets (caller workspace), through the C ABI: same
int64 offsets bit for bit, on two calls in a row, the second with a smaller input (the array keeps
the first call's contents), plus the size limit and the empty input."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _offsets_both(L, nnz):
    R = nnz.shape[0]
    dev = nnz.device
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.full((R + 1,), -7, dtype=torch.int64, device=dev)
    got = torch.full((R + 1,), -9, dtype=torch.int64, device=dev)
    nb = L.smlvm_support_offsets_workspace_bytes(ctypes.c_int64(R))
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
    rc = L.smlvm_support_offsets(ctypes.c_void_p(nnz.data_ptr()), ctypes.c_void_p(ref.data_ptr()),
                                 ctypes.c_int64(R), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nb), st)
    assert rc == 0, rc
    rc = L.smlvm_support_offsets_noalloc(ctypes.c_void_p(nnz.data_ptr()), ctypes.c_void_p(got.data_ptr()),
                                         ctypes.c_int64(R), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return ref, got


def test_noalloc_offsets_match_workspace_offsets(dev):
    from lvmhelpers import _native
    L = _native.lib()
    g = torch.Generator(device=dev).manual_seed(11)
    # first a large call (fills thousands of block sums), then smaller ones that must not see them
    for R in (3_000_001, 70_000, 2049, 2048, 5, 1):
        nnz = torch.randint(0, 40, (R,), device=dev, dtype=torch.int32, generator=g)
        ref, got = _offsets_both(L, nnz)
        assert torch.equal(ref, got), R
        assert int(got[R]) == int(nnz.sum(dtype=torch.int64))
        assert torch.equal(got[:-1], torch.cumsum(nnz.long(), 0) - nnz.long())


def test_noalloc_limits_and_empty(dev):
    from lvmhelpers import _native
    L = _native.lib()
    one = torch.zeros(4, dtype=torch.int64, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = ctypes.c_void_p(one.data_ptr())
    assert L.smlvm_support_offsets_noalloc(P, P, ctypes.c_int64((1 << 24) + 1), st) == -2    # beyond the array
    assert L.smlvm_support_offsets_noalloc(P, None, ctypes.c_int64(4), st) == -1
    assert L.smlvm_support_offsets_noalloc(None, P, ctypes.c_int64(4), st) == -1
    one.fill_(5)
    assert L.smlvm_support_offsets_noalloc(None, P, ctypes.c_int64(0), st) == 0               # empty: offsets[0] = 0
    torch.cuda.synchronize()
    assert int(one[0]) == 0 and int(one[1]) == 5
    # the limit itself is served
    R = 1 << 24
    nnz = torch.ones(R, dtype=torch.int32, device=dev)
    ref, got = _offsets_both(L, nnz)
    assert torch.equal(ref, got) and int(got[R]) == R


def test_noalloc_records_into_a_hip_graph(dev):
    """What it is for: a caller under stream capture (no allocation allowed there)."""
    from lvmhelpers import _native
    L = _native.lib()
    R = 50_000
    nnz = torch.randint(0, 9, (R,), device=dev, dtype=torch.int32)
    off = torch.zeros(R + 1, dtype=torch.int64, device=dev)
    # first call on this device outside the capture (it looks the array's address up once)
    assert L.smlvm_support_offsets_noalloc(ctypes.c_void_p(nnz.data_ptr()), ctypes.c_void_p(off.data_ptr()),
                                           ctypes.c_int64(R), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            rc = L.smlvm_support_offsets_noalloc(ctypes.c_void_p(nnz.data_ptr()), ctypes.c_void_p(off.data_ptr()),
                                                 ctypes.c_int64(R), ctypes.c_void_p(s.cuda_stream))
            assert rc == 0
    torch.cuda.current_stream().wait_stream(s)
    for seed in (1, 2):
        nnz.copy_(torch.randint(0, 9, (R,), device=dev, dtype=torch.int32))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(off[:-1], torch.cumsum(nnz.long(), 0) - nnz.long())
        assert int(off[R]) == int(nnz.sum(dtype=torch.int64))
