"""CPU: the 1.5-entmax oracle (oracle/entmax15_oracle.c) against an independent torch float64
restatement of the published sort/cumsum algorithm, against a bisection solver and against the
optimality conditions.  The third-party package the reference calls is absent: parity unpinned by
the reference, pinned by these."""
import numpy as np
import pytest
import torch


def torch_entmax15(X):
    """Peters et al. 2019, Alg. 2 (entmax 1.0, Entmax15Function), in the dtype of X."""
    X = X / 2
    X = X - X.max(dim=-1, keepdim=True)[0]
    Xs, _ = torch.sort(X, dim=-1, descending=True)
    d = X.shape[-1]
    rho = torch.arange(1, d + 1, dtype=X.dtype).view(1, -1)
    mean = Xs.cumsum(-1) / rho
    mean_sq = (Xs ** 2).cumsum(-1) / rho
    ss = rho * (mean_sq - mean ** 2)
    delta = (1 - ss) / rho
    tau = mean - torch.sqrt(torch.clamp(delta, 0))
    support = (tau <= Xs).sum(-1, keepdim=True)
    return torch.clamp(X - tau.gather(-1, support - 1), min=0) ** 2


def bisect_tau(z):
    lo, hi = -1.0, 0.0
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        if (np.maximum(z - mid, 0.0) ** 2).sum() > 1.0:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def rows(K, sigma, seed, R=40):
    rng = np.random.default_rng(seed)
    x = (sigma * rng.standard_normal((R, K))).astype(np.float32)
    x[0] = 0.5                          # all equal
    if K > 1:
        x[1] = -3.0
        x[1, K // 2] = 4.0              # one dominant logit
        x[2] = np.round(x[2] * 2) / 2   # ties
    return x


@pytest.mark.parametrize("K,sigma", [(1, 1.0), (2, 1.0), (10, 1.0), (256, 1.0), (256, 0.05), (1000, 3.0),
                                     (4096, 1.0), (5000, 0.2)])
def test_forward_matches_restatements(oracle_mod, K, sigma):
    x = rows(K, sigma, K)
    p, nnz, tau = oracle_mod.entmax15_fwd(x)
    ref = torch_entmax15(torch.from_numpy(x).double()).numpy()
    assert np.abs(p - ref).max() <= 5e-7
    assert np.abs(p.sum(1) - 1.0).max() <= 1e-6
    assert np.array_equal(nnz, (p > 0).sum(1))
    z = x.astype(np.float64) / 2
    z -= z.max(1, keepdims=True)
    for r in range(0, x.shape[0], 7):
        assert abs(bisect_tau(z[r]) - tau[r]) <= 2e-7
        # optimality: p_i = max(z_i - tau, 0)^2
        assert np.abs(np.maximum(z[r] - tau[r], 0) ** 2 - p[r]).max() <= 1e-6


def test_backward_and_loss_match_autograd(oracle_mod):
    x = rows(50, 1.0, 3)
    t = np.arange(x.shape[0]) % 50
    xt = torch.from_numpy(x).double().requires_grad_(True)
    P = torch_entmax15(xt)
    dy = torch.from_numpy(np.random.default_rng(1).standard_normal(x.shape)).double()
    (P * dy).sum().backward()
    p, _, _ = oracle_mod.entmax15_fwd(x)
    dx = oracle_mod.entmax15_bwd(dy.numpy().astype(np.float32), p)
    assert np.abs(dx - xt.grad.numpy()).max() <= 2e-5
    # Fenchel-Young loss: value and gradient p - e_y
    loss, grad = oracle_mod.entmax15_loss(x, t)
    pd = P.detach().numpy()
    ref = (1 - (pd * np.sqrt(pd)).sum(1)) / 0.75 + ((pd - np.eye(50)[t]) * x).sum(1)
    assert np.abs(loss - ref).max() <= 2e-5
    assert np.abs(grad - (pd - np.eye(50)[t])).max() <= 1e-6
    assert (loss >= -1e-6).all()
