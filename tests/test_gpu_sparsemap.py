"""GPU parity of kernel 3 (active-set SparseMAP) against the fp64 CPU oracle.

Bar: active set (which configurations, in which order), its length, the iteration count and the
exit status identical; p identical after the fp32 cast (the kernel mirrors the oracle's fp64
operation order, so the fp64 values are the same bits); cached inverse identical; backward within
1e-6 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, n, sigma, seed, kind="normal"):
    rng = np.random.default_rng(seed)
    x = (sigma * rng.standard_normal((B, n))).astype(np.float32)
    if kind == "degenerate":
        x = np.round(x)                        # integers: |x| >= 1 or exactly 0 -> ties everywhere
        x[0] = 0.0
        x[1 % B] = 1.0
        x[2 % B] = -1.0
    init = torch.rand(B, 2 * n, dtype=torch.double,
                      generator=torch.Generator().manual_seed(42)).numpy()
    return x, init


CASES = [(64, 32, 1.0, 0, True), (64, 32, 1.0, 16, True), (64, 32, 3.0, 0, False),
         (64, 32, 3.0, 16, False), (16, 128, 1.0, 0, True), (16, 128, 1.0, 64, True),
         (16, 128, 3.0, 64, False), (8, 5, 1.0, 3, True), (8, 1, 1.0, 0, True),
         (8, 33, 1.0, 5, True), (8, 64, 0.5, 0, False), (8, 100, 2.0, 10, True)]


@pytest.mark.parametrize("kind", ["normal", "degenerate"])
@pytest.mark.parametrize("B,n,sigma,budget,use_init", CASES)
def test_fwd_bwd_match_oracle(dev, oracle_mod, B, n, sigma, budget, use_init, kind):
    from lvmhelpers import _native
    x, init = _inputs(B, n, sigma, 100 * n + budget, kind)
    ref = oracle_mod.sparsemap_fwd(x, init if use_init else None, budget, 300)
    out = _native.sparsemap_fwd(torch.from_numpy(x).to(dev),
                                torch.from_numpy(init).to(dev) if use_init else None, budget, 300)
    alen = out["alen"].cpu().numpy()
    assert np.array_equal(alen, ref["alen"])
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    bits = out["aset_bits"].cpu().numpy().view(np.uint32)
    assert np.array_equal(bits, ref["aset_bits"])
    assert np.array_equal(out["p"].cpu().numpy(), ref["p"])
    inv = out["inv_a"].cpu().numpy()
    for b in range(B):
        m = alen[b]
        assert np.array_equal(inv[b, :m + 1, :m + 1], ref["invA"][b, :m + 1, :m + 1])
    # backward
    rng = np.random.default_rng(5)
    dp = rng.standard_normal((B, n + 1)).astype(np.float32)
    d_ref = oracle_mod.sparsemap_bwd(dp, ref["aset_bits"], ref["alen"], ref["invA"], n)
    d = _native.sparsemap_bwd(torch.from_numpy(dp).to(dev), out["aset_bits"], out["alen"],
                              out["inv_a"], n).cpu().numpy()
    np.testing.assert_allclose(d, d_ref, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(d_ref).max()))


@pytest.mark.parametrize("sigma,budget", [(1.0, 0), (0.3, 64), (1.0, 100)])
def test_tiers_resume_bit_identical(dev, oracle_mod, sigma, budget):
    """n > 64 runs in LDS tiers (storage for 63, 93, 129 active vertices); samples that outgrow a
    tier are parked and resumed by the next launch.  192 samples at 128 bits reach every tier (the
    test asserts that) and must still equal the one-pass CPU oracle bit for bit."""
    from lvmhelpers import _native
    B, n = 192, 128
    x, init = _inputs(B, n, sigma, 4242 + budget)
    ref = oracle_mod.sparsemap_fwd(x, init, budget, 300)
    assert (ref["alen"] > 63).any() and (ref["alen"] <= 63).any()
    if budget == 64:
        assert (ref["alen"] > 93).any()
    out = _native.sparsemap_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(init).to(dev), budget, 300)
    alen = out["alen"].cpu().numpy()
    assert np.array_equal(alen, ref["alen"])
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    assert np.array_equal(out["aset_bits"].cpu().numpy().view(np.uint32), ref["aset_bits"])
    assert np.array_equal(out["p"].cpu().numpy(), ref["p"])
    inv = out["inv_a"].cpu().numpy()
    for b in range(B):
        m = alen[b]
        assert np.array_equal(inv[b, :m + 1, :m + 1], ref["invA"][b, :m + 1, :m + 1])


def test_oversubscribed_batch_starts_in_the_small_tier(dev, oracle_mod):
    """Batches of more than four samples per CU (B > 1024 on an MI355X) start in a 20 KB tier that
    holds 38 active vertices (two waves per SIMD) and are parked / resumed up to three times; the
    result must still equal the one-pass CPU oracle bit for bit (mixed score scales, so that samples
    finish in every tier)."""
    from lvmhelpers import _native
    B, n = 1300, 128
    x, init = _inputs(B, n, 1.0, 777)
    x[::3] *= 3.0           # short active sets: finish in the first tier
    x[1::7] *= 0.6          # long ones
    ref = oracle_mod.sparsemap_fwd(x, init, 0, 300)
    assert (ref["alen"] <= 38).any() and (ref["alen"] > 63).any()
    out = _native.sparsemap_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(init).to(dev), 0, 300)
    alen = out["alen"].cpu().numpy()
    assert np.array_equal(alen, ref["alen"])
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    assert np.array_equal(out["aset_bits"].cpu().numpy().view(np.uint32), ref["aset_bits"])
    assert np.array_equal(out["p"].cpu().numpy(), ref["p"])
    inv = out["inv_a"].cpu().numpy()
    for b in range(0, B, 5):
        m = alen[b]
        assert np.array_equal(inv[b, :m + 1, :m + 1], ref["invA"][b, :m + 1, :m + 1])


@pytest.mark.parametrize("sigma,budget,seq", [(1.0, 0, False), (3.0, 64, False), (1.0, 64, False),
                                              (3.0, 0, False), (1.0, 0, True)])
def test_polish_mode_paths_match_oracle(dev, oracle_mod, sigma, budget, seq):
    """C9-C11 of the oracle (drift monitor, polish mode, re-judged duplicate / singular insertions,
    refined insertion direction) only act on a few % of 128-bit samples, late in their runs: 768
    samples per case, enough for every path to be taken, bit for bit against the CPU oracle."""
    from lvmhelpers import _native
    B, n = 768, 128
    x, init = _inputs(B, n, sigma, 31337 + budget)
    xd, initd = torch.from_numpy(x).to(dev), torch.from_numpy(init).to(dev)
    if seq:
        t = (sigma * np.random.default_rng(5).standard_normal((B, n - 1))).astype(np.float32)
        ref = oracle_mod.sparsemap_seq_fwd(x, t, init, 300)
        out = _native.sparsemap_seq_fwd(xd, torch.from_numpy(t).to(dev), initd, 300)
    else:
        ref = oracle_mod.sparsemap_fwd(x, init, budget, 300)
        out = _native.sparsemap_fwd(xd, initd, budget, 300)
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    alen = out["alen"].cpu().numpy()
    assert np.array_equal(alen, ref["alen"])
    assert np.array_equal(out["aset_bits"].cpu().numpy().view(np.uint32), ref["aset_bits"])
    assert np.array_equal(out["p"].cpu().numpy(), ref["p"])
    inv = out["inv_a"].cpu().numpy()
    for b in range(0, B, 3):
        m = alen[b]
        assert np.array_equal(inv[b, :m + 1, :m + 1], ref["invA"][b, :m + 1, :m + 1])
    # nothing but feasible distributions leaves the solver, whatever the status
    assert np.abs(ref["p64"].sum(-1) - 1).max() <= 1e-9 and ref["p64"].min() >= 0.0


def test_max_iter_is_respected(dev, oracle_mod):
    from lvmhelpers import _native
    x, init = _inputs(8, 64, 1.0, 9)
    for mi in (0, 1, 2, 7, 50):
        ref = oracle_mod.sparsemap_fwd(x, init, 0, mi)
        out = _native.sparsemap_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(init).to(dev), 0, mi)
        assert np.array_equal(out["alen"].cpu().numpy(), ref["alen"])
        assert np.array_equal(out["p"].cpu().numpy(), ref["p"])
        assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])


# Stated bound on the marginals of a sample that leaves at the pass limit (status MAXITER): its
# distribution is the last feasible iterate of a descent method, measured at <= 2.4e-3 from the
# optimum on these inputs (oracle, round 3); 2e-2 is asserted.
MAXITER_MARGINAL_BOUND = 2e-2


@pytest.mark.parametrize("B,sigma,budget", [(1024, 3.0, 0), (1024, 1.0, 0), (1024, 0.3, 0), (1024, 3.0, 64),
                                            (1024, 1.0, 64), (8192, 1.0, 0)])
def test_marginals_closed_form_full_size(dev, oracle_mod, B, sigma, budget):
    """cfg5 shape (1024 samples/GPU at 128 bits, and the whole 8192 batch): EVERY sample -- no filter
    on the exit status -- has p >= 0 and sum p = 1, and M p equals the closed-form marginals to the
    precision of the fp32 output (1e-6) unless the sample left at the pass limit, where the stated
    bound above holds.  Bernoulli factor: every sample converges (C9 of the oracle: drift monitor +
    polish mode; round 2 left 1-4 % at the limit).  Budget factor: >= 98 % converge within the
    reference's 300 passes (structmarg.py:165), the rest is slow convergence with a healthy KKT
    residual (they converge within ~800 passes, tests/test_oracle_sparsemap.py).
    A slice of the batch is compared with the fp64 oracle bit for bit, where the same holds to 1e-9."""
    from lvmhelpers import _native
    n = 128
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, n, device=dev, generator=g) * sigma
    init = torch.rand(B, 2 * n, dtype=torch.double, generator=torch.Generator().manual_seed(3)).to(dev)
    out = _native.sparsemap_fwd(x, init, budget, 300)
    ok = out["status"] == 0  # converged by the KKT test (value <= tau + 1e-9) on an accurate solution
    late = out["status"] == 1
    assert bool((ok | late).all()), "only CONVERGED / MAXITER are expected on these inputs"
    if budget == 0:
        assert bool(ok.all()), f"{int((~ok).sum())} Bernoulli samples did not converge"
    else:
        assert ok.float().mean() >= 0.98
    k = torch.arange(n, device=dev)
    bits = ((out["aset_bits"][..., k // 32] >> (k % 32)) & 1).double()     # [B,cap,n]
    mu = torch.einsum("bc,bcn->bn", out["p"].double(), bits)
    psum = (out["p"].double().sum(-1) - 1).abs()
    assert float(psum.max()) <= 1e-6                      # every sample
    assert float(out["p"].min()) >= 0.0                  # every sample
    assert bool((out["alen"] >= 1).all())
    if budget == 0:
        mu_ref = ((x.double() + 1) / 2).clamp(0, 1)
    else:
        assert bool((bits.sum(-1)[out["p"] > 0] <= budget).all())
        # the marginals are the projection of (x+1)/2 onto {0 <= mu <= 1, sum mu <= budget}
        # (bisection on the multiplier, vectorised over the batch)
        a = (x.double() + 1) / 2
        lo, hi = torch.zeros(B, 1, dtype=torch.double, device=dev), a.max(-1, keepdim=True)[0].clamp(min=0)
        for _ in range(100):
            mid = (lo + hi) / 2
            over = (a - mid).clamp(0, 1).sum(-1, keepdim=True) > budget
            lo, hi = torch.where(over, mid, lo), torch.where(over, hi, mid)
        free = a.clamp(0, 1).sum(-1, keepdim=True) <= budget
        mu_ref = torch.where(free, a.clamp(0, 1), (a - hi).clamp(0, 1))
    err = (mu - mu_ref).abs().max(dim=-1)[0]
    assert float(err[ok].max()) <= 1e-6
    if bool(late.any()):
        assert float(err[late].max()) <= MAXITER_MARGINAL_BOUND
    # the first 96 samples against the oracle: identical, and accurate to 1e-9 in fp64 there
    nb = 96
    ref = oracle_mod.sparsemap_fwd(x[:nb].cpu().numpy(), init[:nb].cpu().numpy(), budget, 300)
    assert np.array_equal(out["status"][:nb].cpu().numpy(), ref["status"])
    assert np.array_equal(out["iters"][:nb].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["alen"][:nb].cpu().numpy(), ref["alen"])
    assert np.array_equal(out["p"][:nb].cpu().numpy(), ref["p"])
    assert np.array_equal(out["aset_bits"][:nb].cpu().numpy().view(np.uint32), ref["aset_bits"])
    inv = out["inv_a"][:nb].cpu().numpy()
    for b in range(nb):
        m = ref["alen"][b]
        assert np.array_equal(inv[b, :m + 1, :m + 1], ref["invA"][b, :m + 1, :m + 1])
    okr = ref["status"] == 0
    assert np.abs(ref["p64"].sum(-1) - 1)[okr].max() <= 1e-9
    if budget == 0:
        bits_r = oracle_mod.unpack_bits(ref["aset_bits"], n).astype(np.float64)
        mu_r = np.einsum("bc,bcn->bn", ref["p64"], bits_r)
        mu_c = np.clip((x[:nb].cpu().numpy().astype(np.float64) + 1) / 2, 0, 1)
        assert np.abs(mu_r - mu_c)[okr].max() <= 1e-9


# ------------------------------------------------------------------ binary sequence factor
def _seq_inputs(B, n, sigma, seed, kind="normal"):
    rng = np.random.default_rng(seed)
    x = (sigma * rng.standard_normal((B, n))).astype(np.float32)
    t = (sigma * rng.standard_normal((B, n - 1))).astype(np.float32)
    if kind == "degenerate":   # integers: ties in Viterbi, affinely dependent vertices (C8 swaps)
        x, t = np.round(x), np.round(t)
        x[0] = 0.0
        t[1 % B] = 0.0
    init = torch.rand(B, 2 * n, dtype=torch.double,
                      generator=torch.Generator().manual_seed(43)).numpy()
    return x, t, init


SEQ_CASES = [(32, 2, 1.0, True), (32, 3, 2.0, False), (32, 8, 1.0, True), (32, 10, 3.0, False),
             (64, 32, 1.0, True), (64, 32, 0.5, False), (16, 33, 1.0, True), (16, 64, 1.0, True),
             (16, 65, 1.0, False), (24, 100, 2.0, True), (24, 128, 1.0, True), (16, 128, 0.5, False)]


@pytest.mark.parametrize("kind", ["normal", "degenerate"])
@pytest.mark.parametrize("B,n,sigma,use_init", SEQ_CASES)
def test_sequence_factor_matches_oracle(dev, oracle_mod, B, n, sigma, use_init, kind):
    """sequence_smap (lvmhelpers/sparsemap.py:47-87,130-148): Viterbi oracle, edge scores in
    Evaluate, vertex swap on singular insertions -- active set, order, iterations, status, p and the
    cached inverse identical to the fp64 oracle; both gradients within 1e-6."""
    from lvmhelpers import _native
    x, t, init = _seq_inputs(B, n, sigma, 31 * n + int(10 * sigma), kind)
    ref = oracle_mod.sparsemap_seq_fwd(x, t, init if use_init else None, max_iter=300)
    out = _native.sparsemap_seq_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev),
                                    torch.from_numpy(init).to(dev) if use_init else None, 300)
    alen = out["alen"].cpu().numpy()
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(alen, ref["alen"])
    assert np.array_equal(out["aset_bits"].cpu().numpy().view(np.uint32), ref["aset_bits"])
    assert np.array_equal(out["p"].cpu().numpy(), ref["p"])
    inv = out["inv_a"].cpu().numpy()
    for b in range(B):
        m = alen[b]
        assert np.array_equal(inv[b, :m + 1, :m + 1], ref["invA"][b, :m + 1, :m + 1])
    rng = np.random.default_rng(6)
    dp = rng.standard_normal((B, n + 1)).astype(np.float32)
    ds_ref, dt_ref = oracle_mod.sparsemap_seq_bwd(dp, ref["aset_bits"], ref["alen"], ref["invA"], n)
    ds, dt = _native.sparsemap_seq_bwd(torch.from_numpy(dp).to(dev), out["aset_bits"], out["alen"],
                                       out["inv_a"], n)
    np.testing.assert_allclose(ds.cpu().numpy(), ds_ref, rtol=1e-6,
                               atol=1e-6 * max(1.0, np.abs(ds_ref).max()))
    np.testing.assert_allclose(dt.cpu().numpy(), dt_ref, rtol=1e-6,
                               atol=1e-6 * max(1.0, np.abs(dt_ref).max()))


def test_sequence_factor_swaps_and_tiers(dev, oracle_mod):
    """192 chains of 128 positions: active sets reach every LDS tier, and the batch contains
    singular insertions (C8); still identical to the one-pass oracle.  With zero edge potentials
    the factor is the Bernoulli factor: marginals clip((x+1)/2, 0, 1)."""
    from lvmhelpers import _native
    B, n = 192, 128
    x, t, init = _seq_inputs(B, n, 1.0, 99)
    t[::2] = np.round(t[::2])
    ref = oracle_mod.sparsemap_seq_fwd(x, t, init, max_iter=300)
    assert (ref["alen"] > 63).any() and (ref["alen"] <= 63).any()
    out = _native.sparsemap_seq_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev),
                                    torch.from_numpy(init).to(dev), 300)
    for key_o, key_r in (("status", "status"), ("iters", "iters"), ("alen", "alen"), ("p", "p")):
        assert np.array_equal(out[key_o].cpu().numpy(), ref[key_r]), key_o
    assert np.array_equal(out["aset_bits"].cpu().numpy().view(np.uint32), ref["aset_bits"])
    # zero edge potentials
    z = np.zeros_like(t)
    out0 = _native.sparsemap_seq_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev),
                                     torch.from_numpy(init).to(dev), 300)
    ok = (out0["status"] == 0).cpu().numpy()
    assert ok.mean() > 0.8
    k = torch.arange(n, device=dev)
    bits = ((out0["aset_bits"][..., k // 32] >> (k % 32)) & 1).double()
    mu = torch.einsum("bc,bcn->bn", out0["p"].double(), bits).cpu().numpy()
    mu_ref = np.clip((x.astype(np.float64) + 1) / 2, 0, 1)
    assert np.abs(mu - mu_ref)[ok].max() <= 1e-6


@pytest.mark.parametrize("n", [4, 6, 9, 12, 16, 24])
def test_sequence_factor_many_swaps(dev, oracle_mod, n):
    """Short chains are where singular insertions (C8 swaps) are frequent: 2048 of them per length,
    a third with integer potentials, a few iteration limits; everything identical to the oracle."""
    from lvmhelpers import _native
    B = 2048
    x, t, init = _seq_inputs(B, n, 1.5, 7 * n)
    x[::3], t[::3] = np.round(x[::3]), np.round(t[::3])
    for mi in (300, 5):
        ref = oracle_mod.sparsemap_seq_fwd(x, t, init, max_iter=mi)
        out = _native.sparsemap_seq_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev),
                                        torch.from_numpy(init).to(dev), mi)
        for key in ("status", "iters", "alen", "p"):
            assert np.array_equal(out[key].cpu().numpy(), ref[key]), (key, mi)
        assert np.array_equal(out["aset_bits"].cpu().numpy().view(np.uint32), ref["aset_bits"])
    assert (ref["iters"] == 5).any()


def test_fuzz_over_shapes_budgets_and_limits(dev, oracle_mod):
    """48 random problems (1 ... 128 bits, any budget, score scales 0.2 ... 4, with and without
    init scores, iteration limits 1 ... 300, integer-valued scores in a quarter of them) and 16
    random chains: everything the solver returns is identical to the oracle."""
    from lvmhelpers import _native
    rng = np.random.default_rng(20260)
    for trial in range(64):
        seq = trial >= 48
        n = int(rng.integers(2 if seq else 1, 129))
        B = int(rng.integers(1, 97))
        sigma = float(rng.choice([0.2, 0.5, 1.0, 2.0, 4.0]))
        budget = 0 if (seq or rng.random() < 0.4) else int(rng.integers(1, n + 1))
        max_iter = int(rng.choice([1, 3, 17, 100, 300]))
        use_init = bool(rng.random() < 0.7)
        x = (sigma * rng.standard_normal((B, n))).astype(np.float32)
        if trial % 4 == 3:
            x = np.round(x)
        init = rng.random((B, 2 * n)) if use_init else None
        it = torch.from_numpy(init).to(dev) if use_init else None
        if seq:
            t = (sigma * rng.standard_normal((B, n - 1))).astype(np.float32)
            ref = oracle_mod.sparsemap_seq_fwd(x, t, init, max_iter=max_iter)
            out = _native.sparsemap_seq_fwd(torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev), it, max_iter)
        else:
            ref = oracle_mod.sparsemap_fwd(x, init, budget, max_iter)
            out = _native.sparsemap_fwd(torch.from_numpy(x).to(dev), it, budget, max_iter)
        tag = (trial, n, B, sigma, budget, max_iter, use_init)
        for key in ("status", "iters", "alen", "p"):
            assert np.array_equal(out[key].cpu().numpy(), ref[key]), (key, tag)
        assert np.array_equal(out["aset_bits"].cpu().numpy().view(np.uint32), ref["aset_bits"]), tag
        inv = out["inv_a"].cpu().numpy()
        for b in range(0, B, 7):
            m = ref["alen"][b]
            assert np.array_equal(inv[b, :m + 1, :m + 1], ref["invA"][b, :m + 1, :m + 1]), tag
