"""Small deterministic encoders / decoders / losses shared by the golden generator and the tests.
Shapes follow BASELINE.json configs 1, 2 (categorical) and 4 (32-bit vectors), scaled down on the
data side so fixtures stay small."""
import numpy as np
import torch
import torch.nn as nn


class CatDecoder(nn.Module):
    """decoder(z [B] long, decoder_input) -> reconstruction [B, D] (cf. MNISTVAE / Receiver)."""

    def __init__(self, K, D):
        super().__init__()
        self.emb = nn.Embedding(K, D)
        self.lin = nn.Linear(D, D)

    def forward(self, z, _decoder_input):
        return self.lin(torch.tanh(self.emb(z)))


class BitDecoder(nn.Module):
    """decoder(Z [N, n] float, *ignored) -> reconstruction [N, D] (cf. CategoricalGenerator)."""

    def __init__(self, n, D):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(n, 16), nn.ReLU(), nn.Linear(16, D))

    def forward(self, Z, *args):
        return self.net(Z)


def cat_loss(_encoder_input, z, decoder_input, decoder_output, labels):
    """(per-row loss [B], logs) like loss_nll of the signal game: logs carry a per-row 'acc'."""
    loss = ((decoder_output - decoder_input) ** 2).sum(-1)
    return loss, {"acc": (z == labels).float()}


def bit_loss(encoder_input, _z, _decoder_input, decoder_output, _labels):
    """(per-row loss [N], {}) like reconstruction_loss of the bit-vector VAE."""
    return ((decoder_output - encoder_input) ** 2).sum(-1), {}


def _seeded(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    return module


class ScoreTable(nn.Module):
    """Encoder stand-in whose scores are a parameter table [B, K]: the scores reaching the hot
    path are then bit-identical on every device (a Linear layer's GEMM rounds differently on CPU
    and GPU, which is enough to move a sparsemax boundary or change an active-set path), and
    d loss / d scores is read off the table's gradient."""

    def __init__(self, B, K, scale=1.0):
        super().__init__()
        self.table = nn.Parameter(torch.zeros(B, K))
        self.scale = scale

    def forward(self, x):
        return self.table


def make_cat_problem(B, D, K, seed):
    enc = _seeded(ScoreTable(B, K), seed)
    with torch.no_grad():
        enc.table.mul_(2.0)
    dec = _seeded(CatDecoder(K, D), seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(B, D, generator=g)
    labels = torch.randint(0, K, (B,), generator=g)
    return enc, dec, x, labels


def make_bit_problem(B, D, n, seed):
    enc = _seeded(ScoreTable(B, n), seed)
    with torch.no_grad():
        enc.table.mul_(2.0)
    dec = _seeded(BitDecoder(n, D), seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(B, D, generator=g)
    return enc, dec, x


def flat_params(*mods):
    return np.concatenate([p.detach().cpu().numpy().ravel() for m in mods for p in m.parameters()])


def flat_grads(*mods):
    return np.concatenate([
        (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().numpy().ravel()
        for m in mods for p in m.parameters()])


def load_flat_params(flat, *mods):
    i = 0
    with torch.no_grad():
        for m in mods:
            for p in m.parameters():
                k = p.numel()
                p.copy_(torch.from_numpy(flat[i:i + k]).view(p.shape))
                i += k
