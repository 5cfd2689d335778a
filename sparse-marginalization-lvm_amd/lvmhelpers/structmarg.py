"""Structured (bit-vector) marginalisation: top-k sparsemax and SparseMAP
(drop-in for lvmhelpers/structmarg.py).

Public surface as in the reference (structmarg.py:23-58, 61-153, 156-202, 205-255).  What runs
where: k-best enumeration, sparsemax, support compaction, bit unpacking, the batched active-set
solver and the weighted loss reduction are gfx950 kernels; encoder/decoder networks and the user
loss stay PyTorch-ROCm ops.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _native
from . import dataparallel
from . import functional as F_


def entropy(p):
    """Same function as lvmhelpers.marg.entropy (the reference keeps a copy: structmarg.py:9-20)."""
    return F_.entropy(p)


def _support_entropy(p, inv_b=1.0):
    """-inv_b sum_{p>0} p log p, no clamping (structmarg.py:51-54, 200), one fused reduction."""
    return F_.support_entropy(p, inv_b)


class TopKSparsemaxWrapper(nn.Module):
    """Scores [B,n] -> (k best bit vectors [B,k,n], sparsemax over their scores [B,k], entropy/B)
    (reference: lvmhelpers/structmarg.py:23-58)."""

    def __init__(self, agent, k=10):
        super().__init__()
        self.agent = agent
        self.k = k

    def forward(self, *args, **kwargs):
        scores = self.agent(*args, **kwargs)
        batch_size, latent_size = scores.shape
        # k-best enumeration stays on the device (the reference round-trips through the host,
        # structmarg.py:42-44)
        bit_vector_z, _ = _native.binary_topk_batched(scores.detach(), self.k)
        topk_scores = torch.einsum("bkj,bj->bk", bit_vector_z, scores)
        distr = F_.sparsemax(topk_scores, dim=-1)
        return bit_vector_z, distr, _support_entropy(distr, 1.0 / batch_size)


class TopKSparsemaxMarg(torch.nn.Module):
    """Training loop of the top-k sparsemax estimator (reference: lvmhelpers/structmarg.py:61-153).

    The reference repeats every input k times and boolean-masks the copies; here the compaction
    kernel emits the (row, slot) lists of the non-zero probabilities in the same row-major order
    and the inputs are gathered by row.
    """

    def __init__(
            self,
            encoder,
            decoder,
            loss_fun,
            encoder_entropy_coeff=0.0,
            decoder_entropy_coeff=0.0):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.loss = loss_fun
        self.encoder_entropy_coeff = encoder_entropy_coeff
        self.decoder_entropy_coeff = decoder_entropy_coeff
        # True: shapes that do not depend on the data and no host synchronisation -- all B * k
        # (row, slot) pairs go through the decoder, the ones with p = 0 behind the real ones as
        # (row 0, slot 0) with weight 0 -- so that a whole step can be captured in a HIP graph
        # (lvmhelpers/graphs.py).  Same loss and gradients.  Training mode only: in eval() the module
        # keeps the data-dependent shapes, whose consumers do not expect padding.
        self.static_shapes = False

    def forward(self, encoder_input, decoder_input, labels):
        bit_vector_z, encoder_probs, encoder_entropy = self.encoder(encoder_input)
        batch_size, k, latent_size = bit_vector_z.shape
        entropy_loss = -(encoder_entropy * self.encoder_entropy_coeff)

        # (row, slot) of every p > 0, row-major == order of the reference's mask (structmarg.py:116)
        if self.static_shapes and self.training:
            off, rows, cols, vals, _ = _native.support_compact_fused(
                encoder_probs.detach(), support_cols=F_.support_cols_of(encoder_probs),
                support_mask=F_.support_mask_of(encoder_probs), static=True)
        else:
            off, rows, cols, vals = _native.support_compact(encoder_probs.detach(), want_vals=True,
                                                            support_cols=F_.support_cols_of(encoder_probs),
                support_mask=F_.support_mask_of(encoder_probs))
        rows_l, cols_l = rows.long(), cols.long()
        encoder_input_sel = encoder_input[rows_l]
        decoder_input_sel = decoder_input[rows_l]
        labels_sel = labels[rows_l]
        # the compaction has the selected probabilities already; their gradient goes back with one scatter
        probs_sel = F_.gather_support(encoder_probs, off, cols, vals)
        bit_vector_z_flat = bit_vector_z[rows_l, cols_l]

        decoder_output = self.decoder(bit_vector_z_flat, decoder_input)
        loss_components, logs = self.loss(
            encoder_input_sel, bit_vector_z_flat, decoder_input_sel, decoder_output, labels_sel)

        loss = F_.weighted_loss(probs_sel, loss_components.to(probs_sel.dtype), 1.0 / batch_size)
        full_loss = loss.mean() + entropy_loss

        for key, v in logs.items():
            if hasattr(v, 'mean'):
                logs[key] = v.mean()
        logs['loss'] = loss.mean().detach()
        logs['encoder_entropy'] = encoder_entropy.detach()
        logs['support'] = (encoder_probs > 0).sum(dim=-1).to(torch.float)
        logs['distr'] = encoder_probs
        logs['loss_output'] = loss_components
        return {'loss': full_loss, 'log': logs}


class SparseMAPWrapper(nn.Module):
    """Scores [B,n] -> SparseMAP distribution over bit vectors, for the whole batch in one launch
    (reference: lvmhelpers/structmarg.py:156-202, which loops over samples in Python).

    Returns ``(sample [N,n], distr [N], entropy/B, idxs: list[int] of length N, support: list[int]
    of length B)`` in batch order, then active-set order, entries with p <= 0 dropped.
    """

    def __init__(self, agent, budget=0, init=False, max_iter=300):
        super().__init__()
        self.agent = agent
        self.budget = budget
        self.init = init
        self.max_iter = max_iter
        self._idx_tensor = None
        # True: `sample` / `distr` keep B * (n + 1) rows whatever the active sets hold (the rows
        # behind the N real ones are sample 0's first configuration with probability 0), `idxs`
        # and `support` are device tensors, and nothing is read back -- what a step captured in a
        # HIP graph needs (lvmhelpers/graphs.py).  With init=True the initial scores live in one
        # buffer that refresh_static() redraws (the same CPU draw as the eager path's, per step).
        self.static_shapes = False
        self._init_buffer = None
        # Data-parallel runs (opt-in): (rank, world, batch_total) of the shard this module sees, or
        # "auto" = ask torch.distributed (equal shards, same CPU seed on every rank -- the caller's
        # responsibility).  None (default) = the reference's draw: torch.rand(2n) per local sample
        # from this process's CPU generator, whatever torch.distributed is used for.
        self.shard = None

    def _init_scores(self, batch_size, latent_size, device):
        # the reference draws torch.rand(2n, dtype=double) from the global CPU generator once per
        # sample (sparsemap.py:25-27); one [B, 2n] draw consumes the same stream.  Sharded over
        # ranks, every rank draws the block of the GLOBAL batch (same seed -> same stream) and
        # keeps its rows, so a sample's init scores -- hence its active set -- do not depend on
        # the world size.
        shard = dataparallel.current_shard(batch_size) if self.shard == "auto" else self.shard
        if shard is None:
            return torch.rand(batch_size, 2 * latent_size, dtype=torch.double).to(device)
        rank, world, batch_total = shard
        init_scores = dataparallel.global_init_scores(batch_total, latent_size, rank, world, device)
        if init_scores.shape[0] != batch_size:
            raise ValueError(f"SparseMAPWrapper.shard {shard} does not describe a batch of {batch_size}")
        return init_scores

    def refresh_static(self):
        """Redraws the static-shape mode's initial scores (one CPU draw per step, as in the eager
        path); a captured step calls it before every replay."""
        if self.init and self._init_buffer is not None:
            b, n2 = self._init_buffer.shape
            self._init_buffer.copy_(self._init_scores(b, n2 // 2, self._init_buffer.device))

    def _forward_static(self, scores):
        batch_size, latent_size = scores.shape
        init_scores = None
        if self.init:
            if self._init_buffer is None or tuple(self._init_buffer.shape) != (batch_size, 2 * latent_size) \
                    or self._init_buffer.device != scores.device:
                self._init_buffer = self._init_scores(batch_size, latent_size, scores.device)
            elif not torch.cuda.is_current_stream_capturing():
                self.refresh_static()
            init_scores = self._init_buffer
        p, aset_bits, alen, _, _ = F_.sparsemap_batched(scores, init_scores, self.budget, self.max_iter)
        off, rows, cols, vals, _ = _native.support_compact_fused(p.detach(), static=True)
        sample = _native.unpack_configs(aset_bits, rows, cols, latent_size)
        distr = F_.gather_support(p, off, cols, vals)
        self._idx_tensor = rows.long()
        support = off[1:] - off[:-1]
        return sample, distr, _support_entropy(distr, 1.0 / batch_size), self._idx_tensor, support

    def forward(self, *args, **kwargs):
        scores = self.agent(*args, **kwargs)
        if self.static_shapes and self.training:   # (evaluation keeps the data-dependent shapes:
            return self._forward_static(scores)     #  its consumers -- importance sampling -- do not expect padding)
        batch_size, latent_size = scores.shape
        init_scores = None
        if self.init:
            init_scores = self._init_scores(batch_size, latent_size, scores.device)
        p, aset_bits, alen, _, _ = F_.sparsemap_batched(
            scores, init_scores, self.budget, self.max_iter)

        # one read-back per step: the row offsets.  N, `support` and `idxs` (row b repeated
        # support[b] times: the compaction is in row-major order) follow from them on the host,
        # while the compaction itself runs on the device.
        # (count + offsets + compaction are one launch; N and the offsets come back in one copy)
        fused = _native.support_compact_fused(p.detach(), want_vals=True, host_offsets=True)
        if fused is None:   # a batch too large for the single-pass bound: separate launches
            off = _native.support_row_offsets(p.detach())
            off_h = off.cpu().numpy()
            _, rows, cols, vals = _native.support_compact(p.detach(), want_vals=True, row_offsets=off,
                                                          total=int(off_h[-1]), fused=False)
        else:
            off, rows, cols, vals, off_h = fused
        counts = np.diff(off_h)
        support = counts.tolist()
        assert min(support) > 0
        idxs = np.repeat(np.arange(batch_size), counts).tolist()
        rows_l = rows.long()
        sample = _native.unpack_configs(aset_bits, rows, cols, latent_size)
        # the compaction wrote the selected probabilities; their gradient goes back with one scatter
        distr = F_.gather_support(p, off, cols, vals)
        self._idx_tensor = rows_l
        return sample, distr, _support_entropy(distr, 1.0 / batch_size), idxs, support


class SparseMAPMarg(torch.nn.Module):
    """Training loop of the SparseMAP estimator (reference: lvmhelpers/structmarg.py:205-255)."""

    def __init__(
            self,
            encoder,
            decoder,
            loss_fun,
            encoder_entropy_coeff=0.0,
            decoder_entropy_coeff=0.0):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.loss = loss_fun
        self.encoder_entropy_coeff = encoder_entropy_coeff
        self.decoder_entropy_coeff = decoder_entropy_coeff

    def forward(self, encoder_input, decoder_input, labels):
        bit_vector_z, encoder_probs, encoder_entropy, idxs, support = \
            self.encoder(encoder_input)
        batch_size = encoder_input.shape[0]
        entropy_loss = -(encoder_entropy * self.encoder_entropy_coeff)

        # gather by the device-side index tensor when the encoder is ours (same values as `idxs`)
        gather = getattr(self.encoder, '_idx_tensor', None)
        if gather is None or gather.numel() != len(idxs):
            gather = idxs
        if torch.is_tensor(support):        # (static-shape mode: per-sample counts on the device)
            support_log = support.to(torch.float)
        else:
            support_log = torch.tensor(support).to(torch.float)

        decoder_output = self.decoder(bit_vector_z)
        loss_components, logs = self.loss(
            encoder_input[gather], bit_vector_z, decoder_input[gather], decoder_output,
            labels[gather])

        loss = F_.weighted_loss(
            encoder_probs, loss_components.to(encoder_probs.dtype), 1.0 / batch_size)
        full_loss = loss + entropy_loss

        for key, v in logs.items():
            if hasattr(v, 'mean'):
                logs[key] = v.mean()
        logs['loss'] = loss.detach()
        logs['encoder_entropy'] = encoder_entropy.detach()
        logs['support'] = support_log
        logs['distr'] = encoder_probs.detach()
        logs['loss_output'] = loss_components.detach()
        logs['idxs'] = idxs
        return {'loss': full_loss, 'log': logs}
