"""Known answers PUBLISHED by the third-party packages the reference delegates to -- the only
outputs of those packages available offline.  The `entmax` package's README (github
deep-spin/entmax, "Example") prints, for x = [-2, 0, 0.5]:
    softmax(x)   = [0.0486, 0.3592, 0.5922]
    sparsemax(x) = [0.0000, 0.2500, 0.7500]
    entmax15(x)  = [0.0000, 0.3260, 0.6740]
(four printed decimals: the tolerance below is half a unit of the last one).  The sparsemax paper
(Martins & Astudillo 2016, sec. 2.2) gives the two-class closed form sparsemax([t, 0])[0] =
clip((t + 1) / 2, 0, 1).  They pin the oracle on CPU and the HIP path on the GPU; everything else
about these two functions stays 'parity unpinned' (see oracle/*.c)."""
import numpy as np
import pytest
import torch

X = np.array([[-2.0, 0.0, 0.5]], np.float32)
SPARSEMAX = np.array([[0.0, 0.25, 0.75]])
ENTMAX15 = np.array([[0.0, 0.3260, 0.6740]])
SOFTMAX = np.array([[0.0486, 0.3592, 0.5922]])


def test_oracle_reproduces_the_entmax_readme(oracle_mod):
    p, s, _, _ = oracle_mod.sparsemax_fwd(X)
    assert np.array_equal(p, SPARSEMAX.astype(np.float32)) and s[0] == 2
    q = oracle_mod.entmax15_fwd(X)[0]
    assert np.abs(q - ENTMAX15).max() <= 5e-5 and q[0, 0] == 0.0
    assert np.abs(torch.softmax(torch.from_numpy(X), -1).numpy() - SOFTMAX).max() <= 5e-5
    t = np.linspace(-3, 3, 25).astype(np.float32)
    two = np.stack([t, np.zeros_like(t)], 1)
    p2 = oracle_mod.sparsemax_fwd(two)[0]
    np.testing.assert_allclose(p2[:, 0], np.clip((t.astype(np.float64) + 1) / 2, 0, 1), atol=1e-7)


@pytest.mark.gpu
def test_gpu_reproduces_the_entmax_readme(dev):
    import entmax   # the shim of this repository, on the HIP kernels
    from lvmhelpers.marg import ExplicitWrapper
    x = torch.from_numpy(X).to(dev)
    assert np.array_equal(entmax.sparsemax(x, dim=-1).cpu().numpy(), SPARSEMAX.astype(np.float32))
    q = entmax.entmax15(x, dim=-1).cpu().numpy()
    assert np.abs(q - ENTMAX15).max() <= 5e-5 and q[0, 0] == 0.0
    # through the reference's class, all three normalisers (lvmhelpers/marg.py:39-54)
    for name, want in (("softmax", SOFTMAX), ("sparsemax", SPARSEMAX), ("entmax", ENTMAX15)):
        _, distr, _ = ExplicitWrapper(torch.nn.Identity(), normalizer=name)(x)
        assert np.abs(distr.cpu().numpy() - want).max() <= 5e-5, name
    t = torch.linspace(-3, 3, 25, device=dev)
    p2 = entmax.sparsemax(torch.stack([t, torch.zeros_like(t)], 1), dim=-1)
    assert float((p2[:, 0].double() - ((t.double() + 1) / 2).clamp(0, 1)).abs().max()) <= 1e-7
