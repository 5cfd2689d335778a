"""Every BASELINE.json model config takes optimisation steps through our lvmhelpers on synthetic
data of the experiment's shape (bench_models.py) and the loss goes down; the fused decoder loss
gives the same step as materialised logits."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _steps(model, batch, n=6, lr=1e-3):
    import bench_models as bm
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    return [float(bm.train_step(model, opt, batch)["loss"]) for _ in range(n)]


@pytest.mark.parametrize("name", ["ssvae", "signal_game", "bitvae_sparsemap", "bitvae_budget",
                                  "bitvae_topk", "bitvae_sparsemap_128"])
def test_config_trains(dev, name):
    import bench_models as bm
    torch.manual_seed(1)
    if name == "ssvae":
        model, batch = bm.make_ssvae(dev, 64)
    elif name == "signal_game":
        model, batch = bm.make_signal_game(dev, 64)
    elif name == "bitvae_sparsemap":
        model, batch = bm.make_bitvae(dev, 64, 32, "sparsemap")
    elif name == "bitvae_budget":
        model, batch = bm.make_bitvae(dev, 64, 32, "sparsemap", budget=16)
    elif name == "bitvae_topk":
        model, batch = bm.make_bitvae(dev, 64, 32, "topksparse")
    else:
        model, batch = bm.make_bitvae(dev, 16, 128, "sparsemap")
    losses = _steps(model, batch)
    assert all(l == l and abs(l) < 1e9 for l in losses)
    assert losses[-1] < losses[0]


def test_fused_decoder_equals_dense_logits(dev):
    import bench_models as bm
    res = []
    for fused in (True, False):
        torch.manual_seed(7)                       # same init scores for the active-set start
        model, batch = bm.make_bitvae(dev, 32, 32, "sparsemap", fused_decoder=fused, seed=3)
        out = model(*batch)
        out["loss"].backward()
        res.append((float(out["loss"]), [p.grad.clone() for p in model.parameters()],
                    out["log"]["idxs"]))
    (l0, g0, i0), (l1, g1, i1) = res
    assert i0 == i1
    assert abs(l0 - l1) <= 3e-6 * abs(l1)
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 5e-5 * max(1.0, float(b.abs().max()))
