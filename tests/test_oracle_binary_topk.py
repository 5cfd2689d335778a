"""CPU: pins the k-best oracle (oracle/binary_topk_oracle.c) against the COMPILED REFERENCE
(oracle/_ref, built from the reference's own binary_topk.h; skipped when it was not built) and
against the committed golden vectors that the compiled reference produced."""
import itertools
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "binary_topk.npz")


def test_golden_vectors_from_compiled_reference():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        cfg, top = oracle.binary_topk_batched(g[f"scores_{i}"], int(g[f"k_{i}"]))
        assert np.array_equal(cfg, g[f"configs_{i}"].astype(np.float32)), i
        assert np.array_equal(top, g[f"top_{i}"]), i
    cfg, top = oracle.binary_topk_batched(np.array([[0.5, -1, 2, 0.1]], np.float32), 5)
    assert np.array_equal(cfg, g["probe_configs"].astype(np.float32))
    assert cfg[0].tolist() == [[1, 0, 1, 1], [1, 0, 1, 0], [0, 0, 1, 1], [0, 0, 1, 0], [1, 1, 1, 1]]


def test_golden_vectors_for_k_beyond_64():
    """binary_topk_wide.npz: the compiled reference at k = 65 ... 1024 (configurations packed)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "binary_topk_wide.npz"))
    assert int(g["n_cases"]) >= 9
    for i in range(int(g["n_cases"])):
        x, k = g[f"scores_{i}"], int(g[f"k_{i}"])
        cfg, top = oracle.binary_topk_batched(x, k)
        want = np.unpackbits(g[f"configs_packed_{i}"], axis=-1)[..., :x.shape[1]].astype(np.float32)
        assert np.array_equal(cfg, want), i
        assert np.array_equal(top, g[f"top_{i}"]), i


@pytest.mark.skipif(oracle.ref_lib() is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("B,n,k", [(50, 32, 10), (20, 128, 10), (30, 6, 64), (10, 512, 40), (40, 3, 8),
                                   (40, 1, 2), (25, 17, 2), (25, 64, 33)])
def test_matches_compiled_reference(B, n, k):
    rng = np.random.default_rng(n * k)
    x = rng.standard_normal((B, n)).astype(np.float32)
    x[::2] = np.round(x[::2])  # ties (including zeros)
    cfg_r, top_r = oracle.ref_binary_topk_batched(x, k)
    cfg, top = oracle.binary_topk_batched(x, k)
    assert np.array_equal(cfg, cfg_r) and np.array_equal(top, top_r)


def test_brute_force_small():
    rng = np.random.default_rng(3)
    n, k = 8, 20
    x = rng.standard_normal((10, n)).astype(np.float32)
    cfg, top = oracle.binary_topk_batched(x, k)
    allc = np.array(list(itertools.product([0, 1], repeat=n)), np.float32)
    for b in range(10):
        sc = np.sort((allc * x[b]).sum(-1).astype(np.float64))[::-1][:k]
        assert np.allclose(top[b], sc, atol=1e-5)
        assert np.allclose((cfg[b] * x[b]).sum(-1), top[b], atol=1e-5)
        assert len({tuple(c) for c in cfg[b]}) == k


def test_ties_prefer_the_off_branch():
    cfg, top = oracle.binary_topk_batched(np.array([[1.0, 1.0]], np.float32), 3)
    assert top[0].tolist() == [2.0, 1.0, 1.0]
    assert cfg[0].tolist() == [[1, 1], [1, 0], [0, 1]]   # binary_topk.h:72: b only if strictly greater


def test_undefined_reference_cases_are_errors():
    with pytest.raises(ValueError):
        oracle.binary_topk_batched(np.zeros((1, 3), np.float32), 9)
    with pytest.raises(ValueError):
        oracle.binary_topk_batched(np.zeros((1, 3), np.float32), 1)
