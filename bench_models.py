"""Synthetic-data stand-ins for the reference's three experiments, wired through OUR lvmhelpers.

The experiment scripts themselves (experiments/*/train.py: PyTorch-Lightning modules, MNIST /
FMNIST / ImageNet-feature loaders) are callers of the hot path and are not rebuilt here; they also
need pytorch_lightning / torchvision / h5py, which this image does not have.  These builders
re-create the *shapes* of the networks those scripts put around the hot path (layer sizes from
experiments/semi_supervised-vae/archs.py, experiments/signal-game/archs.py,
experiments/bit_vector-vae/archs.py + train.py:56-70,374-378) with random-init weights and feed
them random tensors of the data's shape, so that one optimisation step of each BASELINE.json
config can be run and timed on an MI355X: `bench.py` uses them for its `train_steps` block and
tests/test_gpu_train_steps.py checks that every config steps and lowers its loss.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from lvmhelpers import dataparallel
from lvmhelpers.marg import ExplicitWrapper, Marginalizer
from lvmhelpers.structmarg import (SparseMAPMarg, SparseMAPWrapper, TopKSparsemaxMarg,
                                   TopKSparsemaxWrapper)
from lvmhelpers.utils import DeterministicWrapper
from lvmhelpers import fused


def _mlp(sizes, last_act=False):
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2 or last_act:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


# ------------------------------------------------------------------ config 1: semi-supervised VAE
class _GaussianVAE(nn.Module):
    """decoder(z [B] long, image [B,28,28]) -> dict, as MNISTVAE does (archs.py:117-201)."""

    def __init__(self, latent_dim=8, n_classes=10, n_pixels=784):
        super().__init__()
        self.latent_dim, self.n_classes, self.n_pixels = latent_dim, n_classes, n_pixels
        self.enc = _mlp([n_pixels + n_classes, 128, 128, 2 * latent_dim])
        self.dec = _mlp([latent_dim + n_classes, 128, 128, n_pixels])

    def forward(self, z, image):
        onehot = F.one_hot(z, self.n_classes).to(image.dtype)
        h = self.enc(torch.cat((image.reshape(-1, self.n_pixels), onehot), dim=1))
        mean, std = h[:, :self.latent_dim], torch.exp(h[:, self.latent_dim:])
        sample = mean + std * torch.randn_like(mean)
        image_mean = torch.sigmoid(self.dec(torch.cat((sample, onehot), dim=1)))
        return {'latent_means': mean, 'latent_std': std, 'image_mean': image_mean}


def _elbo_loss(image, z, _vae_input, out, labels):
    x = image.reshape(image.shape[0], -1)
    xr = out['image_mean']
    bce = (-x * torch.log(xr + 1e-8) - (1 - x) * torch.log(1 - xr + 1e-8)).sum(1)
    logvar = 2 * torch.log(out['latent_std'])
    kl = ((out['latent_means'] ** 2 + logvar.exp() - 1 - logvar) / 2).sum(1)
    return bce + kl, {'acc': (z == labels).float()}


def make_ssvae(dev, batch=64, seed=0, compact=False):
    torch.manual_seed(seed)
    classifier = _mlp([784, 256, 256, 256, 10])
    model = Marginalizer(ExplicitWrapper(classifier, normalizer='sparsemax'),
                         DeterministicWrapper(_GaussianVAE()), _elbo_loss,
                         encoder_entropy_coeff=1.0, compact=compact).to(dev)
    image = torch.rand(batch, 784, device=dev)
    labels = torch.randint(0, 10, (batch,), device=dev)
    return model, (image, image, labels)


# ------------------------------------------------------------------------- config 2: signal game
class _Sender(nn.Module):
    def __init__(self, feat=4096, emb=256, hidden=512, vocab=256):
        super().__init__()
        self.lin1 = nn.Linear(feat, emb, bias=False)
        self.mix = nn.Parameter(torch.randn(hidden) * 0.1)     # the 1x1 conv: one weight per channel
        self.pool = nn.Parameter(torch.randn(hidden) * 0.1)    # the (hidden x 1) conv
        self.lin4 = nn.Linear(emb, vocab, bias=False)

    def forward(self, x):
        e = self.lin1(x[0])                                     # sender sees the target image only
        h = torch.sigmoid(e.unsqueeze(1) * self.mix.view(1, -1, 1))       # [B, hidden, emb]
        h = torch.sigmoid((h * self.pool.view(1, -1, 1)).sum(1))          # [B, emb]
        return self.lin4(h)


class _Receiver(nn.Module):
    def __init__(self, game_size=16, feat=4096, emb=256, vocab=256):
        super().__init__()
        self.game_size = game_size
        self.lin1 = nn.Linear(feat, emb, bias=False)
        self.lin2 = nn.Embedding(vocab, emb)

    def return_embeddings(self, x):                                 # archs.py:91-105
        return torch.stack([self.lin1(x[i]) for i in range(self.game_size)], dim=1)  # [B, G, emb]

    def forward(self, signal, x):                                   # archs.py:72-89
        embs = self.return_embeddings(x)
        out = torch.bmm(embs, self.lin2(signal).unsqueeze(-1)).squeeze(-1)
        return F.log_softmax(out, dim=1)


def _game_loss(_sender_input, message, _receiver_input, receiver_output, labels):
    nll = F.nll_loss(receiver_output, labels, reduction='none')
    return nll, {'acc': (receiver_output.argmax(1) == labels).float()}


def make_signal_game(dev, batch=64, game_size=16, feat=4096, vocab=256, seed=0, compact=False,
                     hoist=False):
    torch.manual_seed(seed)
    # receiver input is [game_size, B, feat]: the batch axis is 1; the sender input is not needed
    # by the loss, so it is not gathered (batch_dims[0] = None).  hoist: the Receiver embeds the
    # game_size images once per batch and the (row, z) pairs gather the embeddings, so the images
    # are not gathered either (the loss does not look at them).
    model = Marginalizer(ExplicitWrapper(_Sender(feat, 256, 512, vocab), normalizer='sparsemax'),
                         DeterministicWrapper(_Receiver(game_size, feat, 256, vocab)), _game_loss,
                         encoder_entropy_coeff=0.1, compact=compact, hoist=bool(hoist),
                         batch_dims=(None, None if hoist else 1)).to(dev)
    x = torch.randn(game_size, batch, feat, device=dev)       # sender input: x[0] is the target
    labels = torch.randint(0, game_size, (batch,), device=dev)
    recv = x.clone()                                            # receiver sees it at slot labels[b]
    ar = torch.arange(batch, device=dev)
    recv[labels, ar] = x[0, ar]
    recv[0, ar] = x[labels, ar]
    return model, (x, recv, labels)


# --------------------------------------------------------------------- configs 4/5: bit-vector VAE
def make_bitvae(dev, batch=64, bits=32, mode='sparsemap', budget=0, fused_decoder=True, seed=0):
    """inf: 784 -> 128 -> bits; gen: bits -> 128 -> 784*5 -> (rank-5) 256-way logits per pixel
    (train.py:56-70,374-378).  fused_decoder=False materialises the [N,784,256] logits like the
    reference does."""
    torch.manual_seed(seed)
    inf = _mlp([784, 128, bits])
    gen = fused.CategoricalGenerator(_mlp([bits, 128, 784 * 5]), 784, 5, 256)
    if fused_decoder:
        loss = fused.reconstruction_loss
        dec = gen
    else:
        class _Dense(nn.Module):
            def __init__(self, g):
                super().__init__()
                self.g = g

            def forward(self, z, *a):
                return self.g(z, *a).materialize()
        dec = _Dense(gen)
        loss = fused.reconstruction_loss
    if mode == 'sparsemap':
        model = SparseMAPMarg(SparseMAPWrapper(inf, budget=budget, init=True),
                              DeterministicWrapper(dec), loss, encoder_entropy_coeff=1.0)
    else:
        model = TopKSparsemaxMarg(TopKSparsemaxWrapper(inf, k=10), DeterministicWrapper(dec), loss,
                                  encoder_entropy_coeff=1.0)
    model = model.to(dev)
    pixels = torch.randint(0, 256, (batch, 784), device=dev, dtype=torch.uint8)
    x = pixels.float() / 255
    return model, (x, torch.zeros_like(x), pixels)


def train_step(model, optimizer, batch, distributed=False):
    """forward, backward, (one flat-bucket gradient all-reduce), optimiser step."""
    optimizer.zero_grad(set_to_none=True)
    out = model(*batch)
    out['loss'].backward()
    if distributed:
        dataparallel.allreduce_gradients(model.parameters())
    optimizer.step()
    return out
