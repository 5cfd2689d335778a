"""A training step captured as a HIP graph (lvmhelpers/graphs.py) against the same step run eagerly:
same losses, logs and parameters after several steps, for the two structured estimators; and the
static-shape mode on its own (no graph) against the data-dependent shapes."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bitvae(dev, mode, batch, bits, seed=0):
    import bench_models
    torch.manual_seed(seed)
    model, b = bench_models.make_bitvae(dev, batch=batch, bits=bits, mode=mode, seed=seed)
    return model, b


def _batches(dev, batch, n, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    out = []
    for _ in range(n):
        pixels = torch.randint(0, 256, (batch, 784), device=dev, dtype=torch.uint8, generator=g)
        x = pixels.float() / 255
        out.append((x, torch.zeros_like(x), pixels))
    return out


@pytest.mark.parametrize("mode,bits", [("sparsemap", 32), ("topk", 32), ("sparsemap", 12)])
def test_static_shapes_equal_the_data_dependent_ones(dev, mode, bits):
    """static_shapes=True feeds every (sample, slot) pair to the decoder with weight 0 outside the
    support: same loss, same gradients (to rounding: the decoder's GEMMs see another batch size)."""
    model, batch = _bitvae(dev, mode, 16, bits)
    ref = copy.deepcopy(model)
    for m in model.modules():
        if hasattr(m, 'static_shapes'):
            m.static_shapes = True
    torch.manual_seed(11)
    out_ref = ref(*batch)
    out_ref['loss'].backward()
    torch.manual_seed(11)
    out = model(*batch)
    out['loss'].backward()
    assert torch.allclose(out['loss'], out_ref['loss'], rtol=1e-5, atol=1e-5)
    assert torch.allclose(out['log']['encoder_entropy'], out_ref['log']['encoder_entropy'], rtol=1e-5, atol=1e-6)
    assert torch.equal(out['log']['support'].cpu(), out_ref['log']['support'].cpu())
    for p, q in zip(model.parameters(), ref.parameters()):
        scale = max(1.0, float(q.grad.abs().max()))
        assert float((p.grad - q.grad).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("mode,B", [("sparsemap", 16), ("topk", 16), ("sparsemap", 64), ("topk", 64)])
def test_graphed_step_equals_eager_steps(dev, mode, B):
    """(B = 64 is the reference's batch size, and the size from which the decoder's bias gradient --
    a column sum over B * k rows -- needs the semaphores that a captured memset zeroes only once:
    without the repair pass this case turned NaN on the third replay.)"""
    from lvmhelpers.graphs import GraphedStep
    model, batch0 = _bitvae(dev, mode, B, 32)
    # the eager twin runs the same static-shape mode (test above: it equals the data-dependent one
    # on equal parameters).  Over several steps the data-dependent path is no yardstick: parameters
    # that differ by one rounding (another GEMM batch size) move a vertex of probability ~1e-9 in
    # or out of an active set, and SparseMAP's Jacobian jumps with the set -- measured: equal
    # losses, encoder gradients apart by their own size from the second step on.
    ref = copy.deepcopy(model)
    for m in ref.modules():
        if hasattr(m, 'static_shapes'):
            m.static_shapes = True
    # (SGD: Adam divides by the gradient's magnitude, and a weight whose gradient is rounding noise
    # then moves by +-lr depending on the noise's sign -- not a property of the step under test)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
    opt_ref = torch.optim.SGD(ref.parameters(), lr=1e-3, momentum=0.9)
    before = [p.detach().clone() for p in model.parameters()]
    torch.manual_seed(5)
    step = GraphedStep(model, opt, batch0)
    # creating it does not train and leaves the CPU generator where it was
    for p, q in zip(model.parameters(), before):
        assert torch.equal(p, q)
    batches = _batches(dev, B, 4, 3)
    losses, losses_ref = [], []
    for b in batches:
        out = step(*b)
        losses.append(float(out['loss'].detach()))
    torch.manual_seed(5)
    for b in batches:
        opt_ref.zero_grad(set_to_none=True)
        out_ref = ref(*b)
        out_ref['loss'].backward()
        opt_ref.step()
        losses_ref.append(float(out_ref['loss'].detach()))
    for a, b in zip(losses, losses_ref):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (losses, losses_ref)
    moved = 0.0
    for p, q, p0 in zip(model.parameters(), ref.parameters(), before):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-6 * max(1.0, float(q.detach().abs().max()))
        moved = max(moved, float((q.detach() - p0).abs().max()))
    assert moved > 1e-3                                     # (the steps did train)
    assert torch.allclose(out['log']['support'].float().mean().cpu(), out_ref['log']['support'].float().mean().cpu())
    # shapes are fixed at capture
    with pytest.raises(ValueError):
        step(*[t[:B // 2] for t in batches[0]])
    step.release()
    assert not any(getattr(m, 'static_shapes', False) for m in model.modules())


def test_graphed_step_with_adam(dev):
    """Adam needs capturable=True (its step counter on the device); with it the captured step's
    losses follow the eager ones."""
    from lvmhelpers.graphs import GraphedStep
    model, batch = _bitvae(dev, "topk", 8, 16)
    with pytest.raises(ValueError):
        GraphedStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), batch)
    ref = copy.deepcopy(model)
    step = GraphedStep(model, torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True), batch)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-4, capturable=True)
    for b in _batches(dev, 8, 3, 9):
        out = step(*b)
        opt_ref.zero_grad(set_to_none=True)
        out_ref = ref(*b)
        out_ref['loss'].backward()
        opt_ref.step()
        assert abs(float(out['loss'].detach()) - float(out_ref['loss'].detach())) <= 2e-3 * abs(float(out_ref['loss'].detach()))


def test_captured_memsets_are_replaced_by_kernels(dev):
    """lvmhelpers.graphs.capture passes the captured graph through smlvm_graph_memsets_to_kernels:
    a column sum whose launcher zeroes its semaphores with cudaMemsetAsync -- wrong from the second
    replay on with this image's HIP runtime when captured as it is (tools/graph_memset_probe.py) --
    is right on every replay, and the pass reports the node it replaced."""
    from lvmhelpers import graphs
    torch.manual_seed(0)
    x = torch.randn(640, 3920, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            (x * 2).sum(0)
    torch.cuda.current_stream().wait_stream(s)
    box = {}

    def body():
        a = x * 2
        b = torch.relu(a) + 1
        del b
        box['y'] = a.sum(0)
        box['z'] = torch.zeros(1000, device=dev) + a[0, :1000]     # (a fill inside the graph)
    g, replaced = graphs.capture(body)
    assert replaced >= 1
    for _ in range(10):
        x.copy_(torch.randn(640, 3920, device=dev))
        g.replay()
        torch.cuda.synchronize()
        assert float((box['y'] - (x * 2).sum(0)).abs().max()) <= 1e-3
        assert torch.equal(box['z'], (x * 2)[0, :1000])


def _game(dev, batch, hoist, seed=0):
    import bench_models
    return bench_models.make_signal_game(dev, batch=batch, game_size=4, feat=64, vocab=32, seed=seed,
                                         compact=True, hoist=hoist)


@pytest.mark.parametrize("hoist", [False, True])
def test_marginalizer_static_shapes_and_graph(dev, hoist):
    """Marginalizer(compact=True).static_shapes: all B * K (row, latent) pairs, weight 0 outside the
    support -- same loss, logs and gradients as the support-only lists; and the captured step
    follows its eager twin."""
    from lvmhelpers.graphs import GraphedStep
    model, batch = _game(dev, 16, hoist)
    ref = copy.deepcopy(model)
    model.static_shapes = True
    out_ref = ref(*batch)
    out_ref['loss'].backward()
    out = model(*batch)
    out['loss'].backward()
    assert torch.allclose(out['loss'], out_ref['loss'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(out['log']['acc'], out_ref['log']['acc'], rtol=1e-5, atol=1e-6)
    for p, q in zip(model.parameters(), ref.parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 2e-5 * max(1.0, float(q.grad.abs().max()))
    # dense (reference-order) mode cannot be static
    dense, b2 = _game(dev, 8, False)
    dense.compact = False
    dense.static_shapes = True
    with pytest.raises(RuntimeError):
        dense(*b2)
    # captured step == eager static step, over changing batches
    model, batch = _game(dev, 16, hoist, seed=1)
    twin = copy.deepcopy(model)
    twin.static_shapes = True
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9)
    opt_twin = torch.optim.SGD(twin.parameters(), lr=1e-2, momentum=0.9)
    step = GraphedStep(model, opt, batch)
    for seed in range(3):
        _, b = _game(dev, 16, hoist, seed=10 + seed)
        out = step(*b)
        opt_twin.zero_grad(set_to_none=True)
        o2 = twin(*b)
        o2['loss'].backward()
        opt_twin.step()
        assert abs(float(out['loss'].detach()) - float(o2['loss'].detach())) <= 1e-5 * max(1.0, abs(float(o2['loss'].detach())))
    for p, q in zip(model.parameters(), twin.parameters()):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-5 * max(1.0, float(q.detach().abs().max()))


def test_static_cap_and_its_overflow_flag(dev):
    """Marginalizer.static_cap: B * static_cap pairs instead of B * K; same loss while the supports
    fit, and a batch that needs more says so in the log (and a GraphedStep raises when it looks)."""
    from lvmhelpers.graphs import GraphedStep
    model, batch = _game(dev, 16, True)
    ref = copy.deepcopy(model)
    model.static_shapes = True
    with torch.no_grad():
        support = int((ref(*batch)['log']['distr'] > 0).sum(-1).max())
        model.static_cap = support
        out = model(*batch)
        assert not bool(out['log']['support_overflow'])
        assert torch.allclose(out['loss'], ref(*batch)['loss'], rtol=1e-5, atol=1e-6)
        model.static_cap = 1                      # a mean of one pair per row is not enough room
        out = model(*batch)
        assert bool(out['log']['support_overflow'])
        assert bool(torch.isfinite(out['loss']))
    step = GraphedStep(model, torch.optim.SGD(model.parameters(), lr=0.0), batch, check_every=2)
    step(*batch)
    with pytest.raises(RuntimeError):
        step(*batch)


def test_static_shapes_are_for_training_mode_only(dev):
    """After eval() the modules return the data-dependent shapes even with static_shapes on
    (evaluation code walks `idxs` / `support` and does not expect padding)."""
    model, batch = _bitvae(dev, "sparsemap", 8, 16)
    for m in model.modules():
        if hasattr(m, 'static_shapes'):
            m.static_shapes = True
    out = model(*batch)
    assert torch.is_tensor(out['log']['idxs']) and out['log']['loss_output'].numel() == 8 * 17
    model.eval()
    with torch.no_grad():
        out = model(*batch)
    assert isinstance(out['log']['idxs'], list)
    assert out['log']['loss_output'].numel() == len(out['log']['idxs']) == sum(out['log']['support'].int().tolist())


def test_a_batch_of_another_shape_runs_eagerly_on_request(dev):
    """eager_for_other_shapes=True: an epoch's last, smaller batch goes through the same model and
    optimiser eagerly, and the captured step still works afterwards."""
    from lvmhelpers.graphs import GraphedStep
    B = 16
    model, batch0 = _bitvae(dev, "topk", B, 16)
    twin = copy.deepcopy(model)
    for m in twin.modules():
        if hasattr(m, 'static_shapes'):
            m.static_shapes = True
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    opt_twin = torch.optim.SGD(twin.parameters(), lr=1e-3)
    step = GraphedStep(model, opt, batch0, eager_for_other_shapes=True)
    full = _batches(dev, B, 2, 4)
    small = tuple(t[:5] for t in _batches(dev, B, 1, 8)[0])
    for b in (full[0], small, full[1]):
        out = step(*b)
        opt_twin.zero_grad(set_to_none=True)
        o2 = twin(*b)
        o2['loss'].backward()
        opt_twin.step()
        assert abs(float(out['loss'].detach()) - float(o2['loss'].detach())) <= 1e-6 * abs(float(o2['loss'].detach()))
    for p, q in zip(model.parameters(), twin.parameters()):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-6 * max(1.0, float(q.detach().abs().max()))
