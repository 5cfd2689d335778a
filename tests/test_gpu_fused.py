"""GPU tests of the rows next to the hot path: the fused low-rank categorical NLL
(decoder output layer + reconstruction loss of the bit-vector VAE) and the segmented log-sum-exp.

These are floating-point kernels: the reference is the same formula in torch (fp64 on the CPU, and
the reference's own fp32 expression F.linear + F.cross_entropy on the GPU).  Tolerances: loss 2e-6
relative (fp32 sums of 784 terms), gradients 2e-5 of the largest entry."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _problem(N, Fe, C, R, seed, ydtype=torch.uint8, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Fe, R, generator=g) * scale
    w = torch.randn(C, R, generator=g) * scale
    b = torch.randn(C, generator=g)
    y = torch.randint(0, C, (N, Fe), generator=g).to(ydtype)
    gl = torch.randn(N, generator=g)
    return x, w, b, y, gl


def _ref64(x, w, b, y, gl):
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    logits = F.linear(x64, w64, b64)                      # [N,F,C]
    lv = F.cross_entropy(logits.permute(0, 2, 1), y.long(), reduction="none").sum(1)
    (lv * gl.double()).sum().backward()
    return lv.detach(), x64.grad, w64.grad, b64.grad


@pytest.mark.parametrize("N,Fe,C,R,ydtype,scale", [
    (7, 784, 256, 5, torch.uint8, 1.0),       # the bit-vector VAE's decoder shape
    (3, 784, 256, 5, torch.int64, 3.0),       # large logits: the max pass matters
    (5, 33, 10, 1, torch.int32, 1.0),
    (2, 100, 256, 8, torch.uint8, 0.1),
    (1, 1, 2, 3, torch.int64, 1.0),
    (64, 17, 200, 4, torch.uint8, 1.0),
])
def test_lowrank_nll_matches_reference(dev, N, Fe, C, R, ydtype, scale):
    from lvmhelpers import functional as F_
    x, w, b, y, gl = _problem(N, Fe, C, R, 1000 + N, ydtype, scale)
    lv_ref, dx_ref, dw_ref, db_ref = _ref64(x, w, b, y, gl)
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    lv = F_.lowrank_categorical_nll(xd, wd, bd, y.to(dev))
    (lv * gl.to(dev)).sum().backward()
    np.testing.assert_allclose(lv.detach().cpu().double().numpy(), lv_ref.numpy(), rtol=2e-6, atol=1e-5)
    for got, ref in ((xd.grad, dx_ref), (wd.grad, dw_ref), (bd.grad, db_ref)):
        tol = 2e-5 * max(1.0, float(ref.abs().max()))
        assert float((got.cpu().double() - ref).abs().max()) <= tol


@pytest.mark.parametrize("N,Fe,C,R,scale", [(7, 784, 256, 5, 1.0), (3, 784, 256, 5, 3.0), (4, 50, 256, 5, 40.0),
                                            (2, 100, 256, 8, 0.1), (1, 1, 2, 3, 1.0), (64, 17, 200, 4, 1.0)])
def test_forward_dx_direction_equals_the_dx_kernel(dev, N, Fe, C, R, scale):
    """smlvm_lowrank_ce_fwd_dxdir: the forward's extra output times gloss is the dX that the backward
    kernel recomputes from the saved log-sum-exp (incl. logits so large that the bound-shifted sums
    underflow and the element is redone exactly), and the other outputs do not change."""
    from lvmhelpers import _native
    x, w, b, y, gl = _problem(N, Fe, C, R, 77 + N, torch.uint8 if C <= 256 else torch.int64, scale)
    xd, wd, bd, yd, gd = x.to(dev), w.to(dev), b.to(dev), y.to(dev), gl.to(dev)
    loss0, lse0, nll0 = _native.lowrank_ce_fwd(xd, wd, bd, yd)
    loss1, lse1, nll1, dxdir = _native.lowrank_ce_fwd(xd, wd, bd, yd, want_dxdir=True)
    assert torch.equal(loss0, loss1) and torch.equal(lse0, lse1) and torch.equal(nll0, nll1)
    dx_k, _, _ = _native.lowrank_ce_bwd(xd, wd, bd, yd, lse0, gd, want_dx=True, want_dw=False)
    dx_f = dxdir * gd.view(-1, 1, 1)
    _, dx_ref, _, _ = _ref64(x, w, b, y, gl)
    top = max(1.0, float(dx_ref.abs().max()))
    # measured against fp64: 6e-7 / 1.5e-6 / 3.8e-6 of the largest entry at scales 1 / 3 / 40 (the
    # kernel that recomputes softmax = exp(logit - lse) in fp32: 1.4e-6 / 5.6e-6 / 3e-4 -- with logits
    # in the thousands that difference cancels badly; the forward's ratio of two sums does not)
    assert float((dx_f.cpu().double() - dx_ref).abs().max()) <= 1e-5 * top
    if scale <= 3.0:
        assert float((dx_f - dx_k).abs().max()) <= 2e-5 * top


def test_fused_generator_is_a_drop_in(dev):
    """Same state_dict keys and same loss/gradients as the reference pair
    CategoricalGenerator + reconstruction_loss (archs.py:18-29, train.py:361-371), whose dense
    expression is restated here in fp32 torch."""
    from lvmhelpers.fused import CategoricalGenerator, LowRankLogits, reconstruction_loss
    torch.manual_seed(3)
    n_features, out_rank, n_classes, latent, N = 784, 5, 256, 32, 24
    gen = torch.nn.Sequential(torch.nn.Linear(latent, 128), torch.nn.ReLU(),
                              torch.nn.Linear(128, n_features * out_rank))
    model = CategoricalGenerator(gen, n_features, out_rank, n_classes).to(dev)
    assert sorted(model.state_dict().keys()) == sorted(
        ["gen.0.weight", "gen.0.bias", "gen.2.weight", "gen.2.bias", "out.weight", "out.bias"])
    z = (torch.rand(N, latent, device=dev) > 0.5).float()
    labels = torch.randint(0, 256, (N, n_features), device=dev, dtype=torch.uint8)
    out = model(z)
    assert isinstance(out, LowRankLogits) and tuple(out.shape) == (N, n_features, n_classes)
    lv, logs = reconstruction_loss(None, z, None, out, labels)
    assert logs == {} and tuple(lv.shape) == (N,)
    lv.sum().backward()
    grads = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    dense = model(z).materialize()                                # what the reference's forward returns
    lv_ref, _ = reconstruction_loss(None, z, None, dense, labels)  # reference formula (dense branch)
    lv_ref.sum().backward()
    np.testing.assert_allclose(lv.detach().cpu().numpy(), lv_ref.detach().cpu().numpy(), rtol=3e-6)
    for g, p in zip(grads, model.parameters()):
        tol = 5e-5 * max(1.0, float(p.grad.abs().max()))
        assert float((g - p.grad).abs().max()) <= tol


def test_lowrank_nll_full_size_properties(dev):
    """cfg5-sized launch (16384 support rows = 12.8M (row, feature) pairs): permutation of rows
    permutes the losses, and a bias shift leaves every loss unchanged (softmax invariance)."""
    from lvmhelpers import _native
    g = torch.Generator(device=dev).manual_seed(5)
    N, Fe, C, R = 16384, 784, 256, 5
    x = torch.randn(N, Fe, R, device=dev, generator=g)
    w = torch.randn(C, R, device=dev, generator=g)
    b = torch.randn(C, device=dev, generator=g)
    y = torch.randint(0, C, (N, Fe), device=dev, generator=g).to(torch.uint8)
    loss, lse, nll = _native.lowrank_ce_fwd(x, w, b, y)
    perm = torch.randperm(N, device=dev, generator=g)
    loss_p, _, _ = _native.lowrank_ce_fwd(x[perm].contiguous(), w, b, y[perm].contiguous())
    assert torch.equal(loss_p, loss[perm])
    loss_s, _, _ = _native.lowrank_ce_fwd(x, w, b + 2.5, y)
    assert float(((loss_s - loss).abs() / loss).max()) < 2e-6
    assert bool((nll >= 0).all()) and bool(torch.isfinite(lse).all())


@pytest.mark.parametrize("B,maxlen", [(64, 40), (1, 1), (257, 3), (1000, 130)])
def test_segmented_logsumexp(dev, B, maxlen):
    from lvmhelpers.fused import support_logsumexp
    rng = np.random.default_rng(B)
    counts = rng.integers(0 if B > 1 else 1, maxlen + 1, size=B)
    idxs = np.repeat(np.arange(B), counts)
    v = (rng.standard_normal(idxs.size) * 30 - 500).astype(np.float32)
    shift = float(32 * np.log(0.5))
    got = support_logsumexp(torch.from_numpy(v).to(dev), idxs.tolist(), B, shift).cpu().numpy()
    # the reference loop (train.py:325-330)
    tv, ti = torch.from_numpy(v).double(), torch.from_numpy(idxs)
    ref = torch.stack([torch.logsumexp(tv[ti == k] + shift, dim=0) for k in range(B)]).numpy()
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    ok = ~np.isinf(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("ydtype,bad", [(torch.int64, -1), (torch.int64, 10), (torch.int32, 1 << 20), (torch.uint8, 200)])
def test_lowrank_nll_label_out_of_range_is_loud_and_safe(dev, ydtype, bad):
    """A label outside [0, C) (F.cross_entropy raises a device assert for it in the reference,
    train.py:368) never indexes the weights: that row's loss and dx come out NaN, every other row,
    dW and db are what they are without the bad pair's one-hot term."""
    from lvmhelpers import functional as F_
    N, Fe, C, R = 6, 40, 10, 3
    x, w, b, y, gl = _problem(N, Fe, C, R, 5, ydtype)
    y_bad = y.clone()
    y_bad[2, 7] = bad
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    lv = F_.lowrank_categorical_nll(xd, wd, b.to(dev), y_bad.to(dev))
    good = F_.lowrank_categorical_nll(x.to(dev), w.to(dev), b.to(dev), y.to(dev))
    assert bool(torch.isnan(lv[2])) and bool(torch.isfinite(lv[[0, 1, 3, 4, 5]]).all())
    assert torch.equal(lv[[0, 1, 3, 4, 5]], good[[0, 1, 3, 4, 5]])
    gl0 = torch.ones(N, device=dev)
    gl0[2] = 0.0                                # keep the NaN row out of the upstream gradient
    (torch.nan_to_num(lv) * gl0).sum().backward()
    assert bool(torch.isfinite(wd.grad).all())
    assert bool(torch.isfinite(xd.grad[[0, 1, 3, 4, 5]]).all())
