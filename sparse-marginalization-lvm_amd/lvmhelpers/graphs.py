"""A whole training step -- forward, backward, optimiser -- captured once as a HIP graph and replayed.

No counterpart in the reference (its steps run eagerly under PyTorch Lightning).  At the reference's
batch size of 64 a step of the structured estimators is ~60 kernel launches of a few microseconds
each and the GPU idles between them (1.4-2.6 ms per step with the device a few per
cent busy); replayed as one graph the same launches go out back to back.

What makes a step capturable is that nothing in it depends on the data's SHAPE and nothing is read
back: `TopKSparsemaxMarg`, `SparseMAPWrapper` / `SparseMAPMarg` have a `static_shapes` mode for that
(lvmhelpers/structmarg.py) -- every (sample, slot) pair goes through the decoder, the ones outside
the support with weight 0 -- which `GraphedStep` switches on.  The kernels themselves are launched
on the capturing stream like any other op, and their workspaces come from PyTorch's allocator,
which gives a capture its own pool.  The optimiser must keep its state on the device
(`torch.optim.Adam(..., capturable=True)`).

    step = GraphedStep(model, optimizer, first_batch)
    for batch in loader:
        out = step(*batch)          # {'loss': ..., 'log': {...}}: tensors that the next call overwrites

`Marginalizer(compact=True)` has the same mode (all B * K pairs, or B * static_cap of them with an
overflow flag that is checked every `check_every` steps).

The static-shape modes apply in training mode only: after `model.eval()` the modules return the
data-dependent shapes again (evaluation code such as the importance-sampling estimate walks `idxs`
and does not expect padding).

`eager_for_other_shapes=True` runs a batch of another shape (an epoch's last one) eagerly through
the same model and optimiser instead of raising.

Limits: one process (no gradient all-reduce inside the graph); hyper-parameters that are Python
numbers are baked in at capture -- a learning-rate schedule needs the rate as a tensor
(`Adam(lr=torch.tensor(1e-3, device=...), capturable=True)`, updated in place); fixed batch shape (a last, smaller batch needs its own GraphedStep or an eager step); the
static-shape mode pays for B * k (top-k) or B * (n + 1) (SparseMAP) decoder rows instead of the
support's, which is the right trade where a step is launch bound and the wrong one where the
decoder dominates (config 5: 1024 samples of 128 bits per GPU).
"""
import torch


def capture(fn):
    """Runs fn() under stream capture and returns (graph, memset nodes replaced).  The captured
    graph goes through the library's repair pass before it is instantiated: with the HIP runtime
    inside this image's PyTorch wheel a captured memset writes its value on the first replay only
    (an arbitrary word afterwards), and
    PyTorch's reductions zero their semaphores with one (include/sparsemarg.h,
    smlvm_graph_memsets_to_kernels; tools/graph_memset_probe.py shows the failure without it)."""
    from . import _native
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph):
        fn()
    _, replaced = _native.graph_memsets_to_kernels(graph.raw_cuda_graph())
    graph.instantiate()
    return graph, replaced


def _set_static(model, flag):
    touched = []
    for m in model.modules():
        if hasattr(m, 'static_shapes') and m.static_shapes != flag:
            m.static_shapes = flag
            touched.append(m)
    return touched


class GraphedStep:
    """Captures `optimizer.zero_grad(); out = model(*batch); out['loss'].backward(); optimizer.step()`.

    Construction runs `warmup` eager steps on a side stream (lazy initialisations: optimiser state,
    library handles, the allocator's pool) and then puts model, optimiser and the CPU random
    generator back to where they were, so that creating a GraphedStep does not train.
    """

    def __init__(self, model, optimizer, batch, warmup=3, check_every=50, eager_for_other_shapes=False):
        if not all(torch.is_tensor(t) and t.is_cuda for t in batch):
            raise ValueError("GraphedStep: the batch must be a tuple of GPU tensors")
        for group in optimizer.param_groups:
            if 'capturable' in group and not group['capturable']:
                raise ValueError("GraphedStep: the optimiser must be created with capturable=True")
        if not model.training:
            raise ValueError("GraphedStep: the model must be in training mode (model.train())")
        self.model = model
        self.optimizer = optimizer
        self.check_every = check_every
        self.eager_for_other_shapes = eager_for_other_shapes
        self._calls = 0
        self.static_batch = tuple(t.clone() for t in batch)
        self._switched = _set_static(model, True)
        self._refreshers = [m for m in model.modules() if hasattr(m, 'refresh_static')]

        saved_model = [p.detach().clone() for p in model.parameters()]
        saved_buffers = [b.detach().clone() for b in model.buffers()]
        saved_opt = {id(p): {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)}
                     for p, st in optimizer.state.items()}
        cpu_rng = torch.get_rng_state()
        cuda_rng = torch.cuda.get_rng_state(self.static_batch[0].device)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        # the state as it was (in place: the graph will hold these addresses)
        with torch.no_grad():
            for p, q in zip(model.parameters(), saved_model):
                p.copy_(q)
            for b, q in zip(model.buffers(), saved_buffers):
                b.copy_(q)
            # optimiser state the warm-up created starts from zero (= a fresh optimiser's first
            # step), state that was there before goes back to its values
            for p, state in optimizer.state.items():
                before = saved_opt.get(id(p), {})
                for k, v in state.items():
                    if torch.is_tensor(v):
                        if k in before:
                            v.copy_(before[k])
                        else:
                            v.zero_()
        torch.set_rng_state(cpu_rng)
        torch.cuda.set_rng_state(cuda_rng, self.static_batch[0].device)

        optimizer.zero_grad(set_to_none=True)
        self.graph, self.memsets_replaced = capture(lambda: (self._set_out(self._forward_backward()),
                                                             optimizer.step()))

    def _set_out(self, out):
        self.out = out

    def _forward_backward(self):
        out = self.model(*self.static_batch)
        out['loss'].backward()
        return out

    def _eager_step(self):
        for m in self._refreshers:
            m.refresh_static()
        self.optimizer.zero_grad(set_to_none=True)
        out = self._forward_backward()
        self.optimizer.step()
        return out

    def __call__(self, *batch):
        if len(batch) != len(self.static_batch):
            raise ValueError("GraphedStep: the batch has another number of tensors than the captured one")
        for dst, src in zip(self.static_batch, batch):
            if tuple(src.shape) != tuple(dst.shape) or src.dtype != dst.dtype:
                if self.eager_for_other_shapes:     # an epoch's last, smaller batch: the same step, eagerly
                    self.optimizer.zero_grad(set_to_none=False)   # (the graph owns the .grad tensors)
                    out = self.model(*batch)
                    out['loss'].backward()
                    self.optimizer.step()
                    return out
                raise ValueError("GraphedStep: batch shapes and dtypes are fixed at capture "
                                 f"({tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype})")
        for dst, src in zip(self.static_batch, batch):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        for m in self._refreshers:
            m.refresh_static()
        self.graph.replay()
        self._calls += 1
        flag = self.out['log'].get('support_overflow') if isinstance(self.out.get('log'), dict) else None
        if flag is not None and self.check_every and self._calls % self.check_every == 0 and bool(flag):
            raise RuntimeError("GraphedStep: a batch needed more (row, latent) pairs than static_cap "
                               "leaves room for; raise Marginalizer.static_cap")
        return self.out

    def release(self):
        """Back to the data-dependent shapes of the eager path (the graph stays valid but unused)."""
        for m in self._switched:
            m.static_shapes = False
