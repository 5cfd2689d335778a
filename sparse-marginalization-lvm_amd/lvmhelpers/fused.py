"""Callers next to the hot path: decoder-side work over the compacted support.

``CategoricalGenerator`` and ``reconstruction_loss`` are drop-ins for the pair in
experiments/bit_vector-vae/archs.py:18-29 and experiments/bit_vector-vae/train.py:361-371 (same
constructor, same ``state_dict`` keys ``gen.*``, ``out.weight``, ``out.bias``, same loss
signature).  Together they compute the 784 x 256-way cross entropy of every support row without
materialising the ``[N, 784, 256]`` logits (0.8 MB per row): the generator returns a
``LowRankLogits`` handle and the loss hands it to the fused gfx950 kernels.  Any other consumer can
call ``.materialize()`` to get the dense tensor the reference would have produced.

``support_logsumexp`` replaces the per-sample Python loop of compute_importance_sampling
(experiments/bit_vector-vae/train.py:325-330).

``hoisted_decoder`` is the compaction-aware decoder hook of ``Marginalizer(compact=True)``: the part
of a decoder that depends on the decoder input only is computed ONCE per batch and the (row, z)
pairs of the support gather from it.  The signal game's Receiver
(experiments/signal-game/archs.py:72-105) is the case it exists for: its forward embeds all
``game_size`` candidate images of every row for every message z; hoisted, the ``game_size`` images
are embedded once per batch (``return_embeddings``), the N pairs gather their row's ``[game_size,
emb]`` block and one ``bmm`` with the N message embeddings gives the N x game_size scores.
"""
import torch
import torch.nn.functional as F

from . import _native
from . import functional as F_


class LowRankLogits:
    """logits[n, f, c] = x[n, f, :] @ weight[c, :] + bias[c], kept factored."""

    def __init__(self, x, weight, bias):
        self.x, self.weight, self.bias = x, weight, bias

    @property
    def shape(self):
        return torch.Size((self.x.shape[0], self.x.shape[1], self.weight.shape[0]))

    def materialize(self):
        return F.linear(self.x, self.weight, self.bias)


class CategoricalGenerator(torch.nn.Module):
    """experiments/bit_vector-vae/archs.py:18-29 with a lazy output layer."""

    def __init__(self, gen, n_features, out_rank, n_classes):
        super().__init__()
        self.gen = gen
        self.n_features = n_features
        self.out_rank = out_rank
        self.out = torch.nn.Linear(out_rank, n_classes)

    def forward(self, Z, *args):
        X = self.gen(Z)
        X = X.reshape(-1, self.n_features, self.out_rank)
        return LowRankLogits(X, self.out.weight, self.out.bias)


def reconstruction_loss(inf_input, discrete_latent_z, _gen_input, gen_output, true_labels):
    """experiments/bit_vector-vae/train.py:361-371; fused when ``gen_output`` is a LowRankLogits."""
    if isinstance(gen_output, LowRankLogits):
        n, f = gen_output.x.shape[0], gen_output.x.shape[1]
        lv = F_.lowrank_categorical_nll(
            gen_output.x, gen_output.weight, gen_output.bias, true_labels.reshape(n, f))
        return lv, {}
    Xhat_logits = gen_output.permute(0, 2, 1)
    lv = F.cross_entropy(Xhat_logits, true_labels.to(dtype=torch.long), reduction="none")
    return lv.sum(dim=1), {}


def support_logsumexp(values, idxs, batch_size, shift=0.0):
    """[logsumexp(values[idxs == k] + shift) for k in range(batch_size)] as one kernel.

    ``idxs`` is the sorted sample index of every support entry (``log['idxs']`` of SparseMAPMarg,
    a list or a tensor); ``values`` is ``-log['loss_output']``."""
    dev = values.device
    idx = torch.as_tensor(idxs, device=dev, dtype=torch.int64)
    counts = torch.bincount(idx, minlength=batch_size)
    off = torch.zeros(batch_size + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(counts, 0)
    return _native.segmented_logsumexp(values.detach().to(torch.float32).contiguous(), off, shift)


class _EmbedPairs(torch.autograd.Function):
    """weight[idx] for the N (row, symbol) pairs of a step.  Forward: one gather.  Backward:
    onehot(idx)^T @ grad as ONE dense GEMM [V, N] x [N, emb] -- deterministic, and 10-20 us where
    torch's embedding_dense_backward walks the indices feature block by feature block (178 us for
    the signal game's 2048 pairs into a 256 x 256 table, the largest kernel of that step)."""

    @staticmethod
    def forward(ctx, weight, idx):
        ctx.save_for_backward(idx)
        ctx.vocab = weight.shape[0]
        return weight.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        onehot = torch.zeros((idx.shape[0], ctx.vocab), dtype=g.dtype, device=g.device)
        onehot.scatter_(1, idx.unsqueeze(1), 1.0)
        return onehot.t().mm(g), None


_EMBED_PAIRS_MAX = 1 << 26     # one-hot entries (256 MB of fp32) up to which the GEMM form is used


def embed_pairs(embedding, idx):
    """embedding(idx) for a 1-D idx; through _EmbedPairs when `embedding` is a plain nn.Embedding."""
    plain = (isinstance(embedding, torch.nn.Embedding) and embedding.max_norm is None
             and not embedding.scale_grad_by_freq and not embedding.sparse and embedding.padding_idx is None)
    if plain and idx.dim() == 1 and idx.is_cuda and idx.numel() * embedding.num_embeddings <= _EMBED_PAIRS_MAX:
        return _EmbedPairs.apply(embedding.weight, idx)
    return embedding(idx)


class HoistedReceiver:
    """The two-phase view of a module shaped like the signal game's Receiver
    (experiments/signal-game/archs.py:59-105): ``return_embeddings(x) -> [B, game_size, emb]`` and a
    message embedding ``lin2``.  ``decode_pairs`` repeats archs.py:76-89 on gathered rows."""

    def __init__(self, receiver):
        self.receiver = receiver

    def embed_once(self, decoder_input):
        return self.receiver.return_embeddings(decoder_input)

    takes_row_offsets = True      # (Marginalizer then hands the compaction's offsets over as well)

    def decode_pairs(self, z, embeddings, rows, row_offsets=None):
        signal = z
        if signal.dim() == 3:
            signal = signal.squeeze(dim=-1)
        h_s = embed_pairs(self.receiver.lin2, signal)                 # [N, emb]
        if row_offsets is not None and embeddings.dtype == torch.float32 and embeddings.is_cuda \
                and embeddings.shape[0] <= 65535:
            gathered = F_.gather_rows(embeddings.contiguous(), rows, row_offsets)
        else:
            gathered = embeddings.index_select(0, rows)
        out = torch.bmm(gathered, h_s.unsqueeze(-1)).squeeze(-1)
        return F.log_softmax(out, dim=1)


def hoisted_decoder(decoder):
    """The two-phase interface of ``decoder`` (a DeterministicWrapper or a bare module), or None.

    A decoder takes part by defining ``embed_once(decoder_input) -> state`` (any tensor or structure
    whose leading axis is the batch where it is a tensor) and ``decode_pairs(z [N], state, rows [N])
    -> decoder output for the N pairs``; modules with the reference Receiver's members
    (``return_embeddings``, ``lin2``) are adapted as they are, so the unchanged signal-game script
    gets the hoisted path under ``LVMHELPERS_COMPACT=1``."""
    agent = getattr(decoder, 'agent', decoder)
    if callable(getattr(agent, 'embed_once', None)) and callable(getattr(agent, 'decode_pairs', None)):
        return agent
    if callable(getattr(agent, 'return_embeddings', None)) and isinstance(
            getattr(agent, 'lin2', None), torch.nn.Module):
        return HoistedReceiver(agent)
    return None
