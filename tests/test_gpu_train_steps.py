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


@pytest.mark.parametrize("name", ["ssvae", "signal_game", "signal_game_hoisted", "bitvae_sparsemap",
                                  "bitvae_budget", "bitvae_topk", "bitvae_sparsemap_128"])
def test_config_trains(dev, name):
    import bench_models as bm
    torch.manual_seed(1)
    if name == "ssvae":
        model, batch = bm.make_ssvae(dev, 64)
    elif name == "signal_game":
        model, batch = bm.make_signal_game(dev, 64)
    elif name == "signal_game_hoisted":
        model, batch = bm.make_signal_game(dev, 64, compact=True, hoist=True)
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


def test_config5_step_at_the_per_gpu_batch(dev):
    """BASELINE.json configs[4] as ONE step at its per-GPU shape: 1024 samples at 128 bits through
    SparseMAPMarg (batched active-set solver, compaction, bit unpacking, fused rank-5 decoder loss
    over ~50k support rows, weighted reduction, backward through all of it, Adam).  Checks the
    step's own invariants: sum of each sample's distribution is 1, support lists consistent, every
    parameter gets a finite gradient, and three steps lower the loss."""
    import bench_models as bm
    torch.manual_seed(2)
    model, batch = bm.make_bitvae(dev, 1024, 128, "sparsemap", seed=0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.zero_grad(set_to_none=True)
    out = model(*batch)
    out["loss"].backward()
    lg = out["log"]
    idx = torch.tensor(lg["idxs"], device=dev)
    N = idx.numel()
    assert lg["distr"].shape == (N,) and lg["loss_output"].shape == (N,)
    assert lg["support"].shape == (1024,) and int(lg["support"].sum()) == N
    assert bool((idx[1:] >= idx[:-1]).all()) and int(idx[0]) == 0 and int(idx[-1]) == 1023
    assert torch.equal(torch.bincount(idx, minlength=1024).float().cpu(), lg["support"].cpu())
    sums = torch.zeros(1024, device=dev, dtype=torch.float64).index_add_(0, idx, lg["distr"].double())
    assert float((sums - 1).abs().max()) <= 1e-6          # every sample: the fp32 precision of p
    assert bool((lg["distr"] > 0).all())
    for p in model.parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all())
    opt.step()
    losses = [float(out["loss"])] + [float(bm.train_step(model, opt, batch)["loss"]) for _ in range(3)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]
