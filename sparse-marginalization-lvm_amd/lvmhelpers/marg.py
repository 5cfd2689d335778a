"""Exact marginalisation over a categorical latent (drop-in for lvmhelpers/marg.py).

Same public surface as the reference (marg.py:6,31-54,57-132); the arithmetic of the hot path --
sparsemax forward/backward, entropy, the probability-weighted loss reduction -- runs in the
gfx950 kernels behind ``include/sparsemarg.h``.
"""
import os

import torch
import torch.nn as nn

from . import _native
from . import functional as F_
from . import fused


def entropy(p: torch.Tensor):
    """Shannon entropy of each row of ``p`` [batch, categories], exact zeros allowed
    (reference: lvmhelpers/marg.py:6-28).  Returns [batch]."""
    return F_.entropy(p)


def _entmax15(X, dim=-1):
    return F_.entmax15(X, dim=dim)   # gfx950 kernel (smlvm_entmax15_fwd/_bwd)


class ExplicitWrapper(nn.Module):
    """Wraps a scoring network: ``forward`` returns ``(argmax sample [B], distr [B,K],
    entropy [B])`` (reference: lvmhelpers/marg.py:31-54)."""

    def __init__(self, agent, normalizer='entmax'):
        super().__init__()
        self.agent = agent
        # KeyError for an unknown name, like the reference's dict lookup (marg.py:43-47)
        self.normalizer = {
            'softmax': torch.softmax,
            'sparsemax': F_.sparsemax,
            'entmax': _entmax15}[normalizer]
        self._fused = normalizer == 'sparsemax'

    def forward(self, *args, **kwargs):
        scores = self.agent(*args, **kwargs)
        if self._fused and scores.dim() == 2:
            # one kernel: sparsemax + entropy epilogue
            distr, entropy_distr, _ = F_.sparsemax_with_entropy(scores)
        else:
            distr = self.normalizer(scores, dim=-1)
            entropy_distr = entropy(distr)
        sample = scores.argmax(dim=-1)
        return sample, distr, entropy_distr


class Marginalizer(torch.nn.Module):
    """Training loop of the exact-marginalisation estimator (reference: lvmhelpers/marg.py:57-132).

    ``encoder`` is an ExplicitWrapper, ``decoder`` a DeterministicWrapper, ``loss_fun(encoder_input,
    z, decoder_input, decoder_output, labels) -> (per-row loss [B], dict)``.

    The decoder is run on the full batch once per latent value that is live for at least one row,
    exactly like the reference, but the live set is read back with ONE host sync instead of one per
    category (marg.py:87), and ``sum_z p(z|x) * loss_z`` is the fused weighted-reduction kernel.

    ``compact=True`` (keyword only here; or ``LVMHELPERS_COMPACT=1`` in the environment for
    unchanged scripts) is the support-only mode the compaction kernel exists for: the (row, latent)
    pairs with p > 0 are listed, the inputs are gathered by row along ``batch_dims = (encoder_input
    dim, decoder_input dim)`` (``None`` = hand the tensor over as it is; default (0, 0), or
    ``LVMHELPERS_BATCH_DIMS="none,1"`` in the environment) and the decoder and the loss
    run ONCE over N = sum_b |supp_b| rows instead of B rows for each of the |union of supports|
    live columns.  Loss, gradients and every log value are the same sums in a different order; a
    decoder that draws random numbers (the SSVAE's) consumes the generator differently.

    ``hoist`` (compact mode; default: whenever the decoder offers it, see ``fused.hoisted_decoder``):
    the decoder's input-only part runs once per batch on the UNgathered ``decoder_input`` and the
    pairs gather from its result -- for the signal game's Receiver
    (experiments/signal-game/archs.py:72-105) that is 16 image embeddings per row instead of 16 per
    (row, z) pair, and the [game_size, N, feat] gather of the images disappears.  The loss is then
    handed ``decoder_input`` gathered only if ``batch_dims[1]`` is not None.
    """

    def __init__(
            self,
            encoder,
            decoder,
            loss_fun,
            encoder_entropy_coeff=0.0,
            decoder_entropy_coeff=0.0,
            *,
            compact=None,
            batch_dims=None,
            hoist=None):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.loss = loss_fun
        self.encoder_entropy_coeff = encoder_entropy_coeff
        self.decoder_entropy_coeff = decoder_entropy_coeff
        self.compact = (os.environ.get("LVMHELPERS_COMPACT") == "1") if compact is None else bool(compact)
        if batch_dims is None:   # unchanged scripts: LVMHELPERS_BATCH_DIMS="none,1" (signal game)
            spec = os.environ.get("LVMHELPERS_BATCH_DIMS", "0,0").split(",")
            if len(spec) != 2:
                raise ValueError("LVMHELPERS_BATCH_DIMS: two comma-separated entries (an axis or 'none')")
            batch_dims = tuple(None if e.strip().lower() == "none" else int(e) for e in spec)
        self.batch_dims = batch_dims
        self.hoist = hoist
        # compact mode only.  True: all B * K (row, latent) pairs go through the decoder, the ones
        # with p = 0 behind the real ones as (row 0, latent 0) with weight 0; shapes then do not
        # depend on the data and nothing is read back, so a whole step can be captured in a HIP
        # graph (lvmhelpers/graphs.py).  Worth it for small K (the SSVAE's 10 classes) or a hoisted
        # decoder whose per-pair part is cheap (the signal game's Receiver).
        self.static_shapes = False
        # static mode: room for this many (row, latent) pairs per row, B * static_cap in all, instead
        # of K (None).  A batch whose supports need more loses the excess pairs -- the step's log
        # then carries 'support_overflow' = True (a device tensor; GraphedStep looks at it every
        # `check_every` steps and raises).  For large K with small supports: the signal game's 256
        # symbols at a support of ~6.
        self.static_cap = None

    def _forward_compact(self, encoder_input, decoder_input, labels, discrete_latent_z,
                         encoder_probs, encoder_entropy):
        batch_size, latent_size = encoder_probs.shape
        overflow = None
        if self.static_shapes and self.training:
            n_cap = None if self.static_cap is None else batch_size * min(latent_size, int(self.static_cap))
            off, rows, cols, vals, n_total = _native.support_compact_fused(
                encoder_probs.detach(), support_cols=F_.support_cols_of(encoder_probs),
                support_mask=F_.support_mask_of(encoder_probs), static=True,
                n_cap=n_cap)
            if n_cap is not None:
                overflow = (n_total > n_cap).reshape(())
        else:
            off, rows, cols, vals = _native.support_compact(encoder_probs.detach(), want_vals=True,
                                                            support_cols=F_.support_cols_of(encoder_probs),
                support_mask=F_.support_mask_of(encoder_probs))
        rows_l, cols_l = rows.long(), cols.long()
        if rows_l.numel() == 0:
            raise RuntimeError("Marginalizer: the encoder distribution has no live category")

        def take(t, dim):
            if dim is None:
                return t
            if isinstance(t, (list, tuple)):   # a list of per-slot tensors counts as axis 0
                if dim < 1:
                    raise ValueError(
                        "Marginalizer(compact=True): a list/tuple input counts as axis 0 (one entry "
                        "per slot), so its batch axis must be >= 1 -- e.g. batch_dims=(None, 1) or "
                        "LVMHELPERS_BATCH_DIMS=\"none,1\" for the signal game's [game_size][B, feat] "
                        f"receiver input; got batch axis {dim}")
                return type(t)(take(e, dim - 1) for e in t)
            if t.shape[dim] != batch_size:   # (a gather with out-of-range rows faults the GPU)
                raise ValueError(
                    f"Marginalizer(compact=True): axis {dim} of an input of shape {tuple(t.shape)} is "
                    f"not the batch axis (batch size {batch_size}); set batch_dims / LVMHELPERS_BATCH_DIMS")
            return t.index_select(dim, rows_l)

        encoder_input_sel = take(encoder_input, self.batch_dims[0])
        two_phase = None if self.hoist is False else fused.hoisted_decoder(self.decoder)
        if self.hoist and two_phase is None:
            raise TypeError("Marginalizer(hoist=True): the decoder defines neither embed_once/"
                            "decode_pairs nor return_embeddings/lin2")
        if two_phase is not None:
            if getattr(two_phase, 'takes_row_offsets', False):
                decoder_output = two_phase.decode_pairs(cols_l, two_phase.embed_once(decoder_input), rows_l,
                                                        row_offsets=off)
            else:
                decoder_output = two_phase.decode_pairs(cols_l, two_phase.embed_once(decoder_input), rows_l)
            decoder_input_sel = take(decoder_input, self.batch_dims[1])
        else:
            decoder_input_sel = take(decoder_input, self.batch_dims[1])
            decoder_output = self.decoder(cols_l, decoder_input_sel)
        loss_n, logs = self.loss(
            encoder_input_sel, discrete_latent_z[rows_l], decoder_input_sel, decoder_output,
            labels[rows_l])
        # (the compaction wrote the selected probabilities; one scatter takes their gradient back)
        probs_sel = F_.gather_support(encoder_probs, off, cols, vals)
        loss_mean = F_.weighted_loss(probs_sel, loss_n.to(encoder_probs.dtype), 1.0 / batch_size)
        for k, v in logs.items():
            if hasattr(v, 'mean'):   # sum_z (p[:, z] * v_z).mean() of the reference, marg.py:107-113
                logs[k] = (probs_sel * v).sum() / batch_size
        entropy_loss = -(encoder_entropy.mean() * self.encoder_entropy_coeff)
        logs['loss'] = loss_mean
        logs['encoder_entropy'] = encoder_entropy.mean()
        logs['support'] = (encoder_probs != 0).sum(-1).to(torch.float).mean()
        logs['distr'] = encoder_probs
        if overflow is not None:
            logs['support_overflow'] = overflow
        return {'loss': loss_mean + entropy_loss.mean(), 'log': logs}

    def forward(self, encoder_input, decoder_input, labels):
        discrete_latent_z, encoder_probs, encoder_entropy = self.encoder(encoder_input)
        batch_size, latent_size = encoder_probs.shape
        if self.compact:
            return self._forward_compact(encoder_input, decoder_input, labels, discrete_latent_z,
                                         encoder_probs, encoder_entropy)
        if self.static_shapes and self.training:
            raise RuntimeError("Marginalizer: static_shapes needs compact=True (the reference-order "
                               "column loop reads the live columns back on the host)")
        entropy_loss = -(encoder_entropy.mean() * self.encoder_entropy_coeff)

        # columns whose probability mass is non-zero for some row (marg.py:87), one sync
        live = (encoder_probs.detach().sum(dim=0) != 0).nonzero().flatten().tolist()

        column_losses = []
        logs, logs_global = None, None
        for z in live:
            z_batch = torch.full((batch_size,), z, dtype=torch.long, device=encoder_probs.device)
            decoder_output = self.decoder(z_batch, decoder_input)
            # NB the reference hands the ARGMAX sample to the loss, not z (marg.py:98-100)
            loss_z, logs = self.loss(
                encoder_input, discrete_latent_z, decoder_input, decoder_output, labels)
            column_losses.append(loss_z)
            if not logs_global:
                logs_global = {k: 0.0 for k in logs.keys()}
            for k, v in logs.items():
                if hasattr(v, 'mean'):
                    logs_global[k] += (encoder_probs[:, z] * v).mean()
        if logs is None:
            raise RuntimeError("Marginalizer: the encoder distribution has no live category")
        for k, v in logs.items():
            if hasattr(v, 'mean'):
                logs[k] = logs_global[k]

        # mean_b sum_z p[b,z] * loss[b,z] over the live columns (the others contribute 0)
        losses = torch.stack(column_losses, dim=1)                      # [B, |live|]
        probs_live = encoder_probs[:, live] if len(live) < latent_size else encoder_probs
        loss_mean = F_.weighted_loss(
            probs_live.reshape(-1), losses.reshape(-1).to(encoder_probs.dtype), 1.0 / batch_size)
        full_loss = loss_mean + entropy_loss.mean()

        logs['loss'] = loss_mean
        logs['encoder_entropy'] = encoder_entropy.mean()
        logs['support'] = (encoder_probs != 0).sum(-1).to(torch.float).mean()
        logs['distr'] = encoder_probs
        return {'loss': full_loss, 'log': logs}
