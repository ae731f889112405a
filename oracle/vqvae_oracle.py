"""CPU oracle for the VQ-VAE-2 stage-1 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

This is an independent, functional (no nn.Module) restatement on plain
PyTorch-CPU fp32 of what the reference computes on this path.  Each function
cites the reference lines it follows (paths relative to /root/reference):

  quantize_forward  <- vqvae.py:42-75   (Quantize.forward), 77-78 (embed_code)
  resblock          <- vqvae.py:81-96
  encoder           <- vqvae.py:99-127
  decoder           <- vqvae.py:130-166
  vqvae_encode      <- vqvae.py:223-240
  vqvae_forward     <- vqvae.py:216-221
  vqvae_decode_code <- vqvae.py:251-259 (upstream semantics; the fork's call is broken)
  stage1_loss       <- train_vqvae.py:31,34,83-85
  adam_step         <- torch.optim.Adam defaults used at train_vqvae.py:185
  CycleSchedule     <- scheduler.py:221-320

Parity pin: oracle/make_golden.py runs the *reference's own* vqvae.py in the
build container on the same generated weights/inputs and stores its outputs in
tests/golden/*.npz; tests/test_oracle_golden.py checks this file against them.
"""
import math
import zlib
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from . import rng


# --------------------------------------------------------------------------
# configuration + state_dict layout (vqvae.py:169-212)
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class VQVAEConfig:
    in_channel: int = 3
    channel: int = 128
    n_res_block: int = 2
    n_res_channel: int = 32
    embed_dim: int = 64
    n_embed: int = 512
    decay: float = 0.99  # accepted but never forwarded by the reference (vqvae.py:185,190)
    eps: float = 1e-5


TINY = VQVAEConfig(channel=32, n_res_block=1, n_res_channel=8, embed_dim=16, n_embed=64)
DEFAULT = VQVAEConfig()


def _encoder_spec(prefix, in_ch, ch, n_res, n_res_ch, stride):
    spec = []
    if stride == 4:
        spec += [(f"{prefix}.blocks.0", "conv", (ch // 2, in_ch, 4, 4)),
                 (f"{prefix}.blocks.2", "conv", (ch, ch // 2, 4, 4)),
                 (f"{prefix}.blocks.4", "conv", (ch, ch, 3, 3))]
        base = 5
    elif stride == 2:
        spec += [(f"{prefix}.blocks.0", "conv", (ch // 2, in_ch, 4, 4)),
                 (f"{prefix}.blocks.2", "conv", (ch, ch // 2, 3, 3))]
        base = 3
    else:
        raise ValueError("stride must be 2 or 4")
    for i in range(n_res):
        spec += [(f"{prefix}.blocks.{base + i}.conv.1", "conv", (n_res_ch, ch, 3, 3)),
                 (f"{prefix}.blocks.{base + i}.conv.3", "conv", (ch, n_res_ch, 1, 1))]
    return spec


def _decoder_spec(prefix, in_ch, out_ch, ch, n_res, n_res_ch, stride):
    spec = [(f"{prefix}.blocks.0", "conv", (ch, in_ch, 3, 3))]
    for i in range(n_res):
        spec += [(f"{prefix}.blocks.{1 + i}.conv.1", "conv", (n_res_ch, ch, 3, 3)),
                 (f"{prefix}.blocks.{1 + i}.conv.3", "conv", (ch, n_res_ch, 1, 1))]
    t = n_res + 2
    if stride == 4:
        spec += [(f"{prefix}.blocks.{t}", "convT", (ch, ch // 2, 4, 4)),
                 (f"{prefix}.blocks.{t + 2}", "convT", (ch // 2, out_ch, 4, 4))]
    elif stride == 2:
        spec += [(f"{prefix}.blocks.{t}", "convT", (ch, out_ch, 4, 4))]
    else:
        raise ValueError("stride must be 2 or 4")
    return spec


def layer_spec(cfg: VQVAEConfig):
    """[(module prefix, kind, weight shape)] in the reference's registration order."""
    c, e = cfg.channel, cfg.embed_dim
    spec = []
    spec += _encoder_spec("enc_b", cfg.in_channel, c, cfg.n_res_block, cfg.n_res_channel, 4)
    spec += _encoder_spec("enc_t", c, c, cfg.n_res_block, cfg.n_res_channel, 2)
    spec += [("quantize_conv_t", "conv", (e, c, 1, 1)), ("quantize_t", "vq", (e, cfg.n_embed))]
    spec += _decoder_spec("dec_t", e, e, c, cfg.n_res_block, cfg.n_res_channel, 2)
    spec += [("quantize_conv_b", "conv", (e, e + c, 1, 1)), ("quantize_b", "vq", (e, cfg.n_embed))]
    spec += [("upsample_t", "convT", (e, e, 4, 4))]
    spec += _decoder_spec("dec", e + e, cfg.in_channel, c, cfg.n_res_block, cfg.n_res_channel, 4)
    # dead decoder kept for state_dict parity (vqvae.py:203-210)
    spec += _decoder_spec("dec_ir", e + e, 1, c, cfg.n_res_block + 2, cfg.n_res_channel, 4)
    return spec


def state_spec(cfg: VQVAEConfig):
    """Ordered {key: shape} equal to reference VQVAE(**cfg).state_dict() keys/shapes."""
    out = {}
    for name, kind, shape in layer_spec(cfg):
        if kind == "vq":
            out[f"{name}.embed"] = shape
            out[f"{name}.cluster_size"] = (shape[1],)
            out[f"{name}.embed_avg"] = shape
        else:
            out[f"{name}.weight"] = shape
            out[f"{name}.bias"] = (shape[0],) if kind == "conv" else (shape[1],)
    return out


def make_state(cfg: VQVAEConfig, seed=1234):
    """Deterministic synthetic state_dict (torch CPU fp32) from the counter RNG.

    Conv/convT weights and biases ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in)) (same
    scale as torch's default init); codebooks ~ N(0,1) (vqvae.py:37),
    cluster_size = 0, embed_avg = embed.clone() (vqvae.py:38-40).
    """
    st = {}
    for name, kind, shape in layer_spec(cfg):
        if kind == "vq":
            emb = rng.normal(seed, name + ".embed", shape)
            st[f"{name}.embed"] = torch.from_numpy(emb.copy())
            st[f"{name}.cluster_size"] = torch.zeros(shape[1])
            st[f"{name}.embed_avg"] = torch.from_numpy(emb.copy())
        else:
            if kind == "conv":
                fan_in = shape[1] * shape[2] * shape[3]
                nb = shape[0]
            else:  # convT weight is (Cin, Cout, kh, kw); torch computes fan_in from dim 1
                fan_in = shape[1] * shape[2] * shape[3]
                nb = shape[1]
            b = 1.0 / math.sqrt(fan_in)
            st[f"{name}.weight"] = torch.from_numpy(rng.uniform(seed, name + ".weight", shape, -b, b))
            st[f"{name}.bias"] = torch.from_numpy(rng.uniform(seed, name + ".bias", (nb,), -b, b))
    return st


def make_images(batch, size, seed=1234, rank=0, in_channel=3):
    """fp32 NCHW i.i.d. N(0,1) images; stream depends on rank."""
    return torch.from_numpy(rng.normal(seed + rank, "images", (batch, in_channel, size, size)))


BUFFER_SUFFIXES = (".embed", ".cluster_size", ".embed_avg")


def is_buffer(key):
    return key.endswith(BUFFER_SUFFIXES)


def is_live_param(key):
    """Trainable AND used by forward (dec_ir never gets a gradient)."""
    return (not is_buffer(key)) and (not key.startswith("dec_ir."))


# --------------------------------------------------------------------------
# Quantize (vqvae.py:28-78)
# --------------------------------------------------------------------------
def quantize_distances(flatten, embed):
    """vqvae.py:44-48: ||x||^2 - 2 x@E + ||e||^2, evaluated in that order."""
    return (flatten.pow(2).sum(1, keepdim=True)
            - 2 * flatten @ embed
            + embed.pow(2).sum(0, keepdim=True))


def quantize_forward(x, embed, cluster_size, embed_avg, training, decay=0.99, eps=1e-5,
                     all_reduce=None):
    """x [...,D] -> (ste_out [...,D], diff 0-dim, idx [...] int64).

    In training mode the three buffers are updated IN PLACE after the gather
    (vqvae.py:52 precedes 54-70), so the output uses the pre-update codebook.
    `all_reduce(t)` sums t across ranks in place (vqvae.py:58-59); None = world 1.
    """
    dim, n_embed = embed.shape
    flatten = x.reshape(-1, dim)
    dist = quantize_distances(flatten, embed)
    _, embed_ind = (-dist).max(1)  # first maximal index on ties (vqvae.py:49)
    embed_onehot = F.one_hot(embed_ind, n_embed).type(flatten.dtype)
    embed_ind = embed_ind.view(*x.shape[:-1])
    quantize = F.embedding(embed_ind, embed.transpose(0, 1))  # vqvae.py:77-78
    if training:
        with torch.no_grad():
            onehot_sum = embed_onehot.sum(0)
            embed_sum = flatten.detach().transpose(0, 1) @ embed_onehot
            if all_reduce is not None:
                all_reduce(onehot_sum)
                all_reduce(embed_sum)
            ema_update_(embed, cluster_size, embed_avg, onehot_sum, embed_sum, decay, eps)
    diff = (quantize.detach() - x).pow(2).mean()
    quantize = x + (quantize - x).detach()
    return quantize, diff, embed_ind


def ema_update_(embed, cluster_size, embed_avg, onehot_sum, embed_sum, decay=0.99, eps=1e-5):
    """vqvae.py:61-70 (in place)."""
    n_embed = embed.shape[1]
    cluster_size.mul_(decay).add_(onehot_sum, alpha=1 - decay)
    embed_avg.mul_(decay).add_(embed_sum, alpha=1 - decay)
    n = cluster_size.sum()
    cs = (cluster_size + eps) / (n + n_embed * eps) * n
    embed.copy_(embed_avg / cs.unsqueeze(0))


def quantize_stats(x, idx, n_embed):
    """(counts [K], sums [D,K]) -- what vqvae.py:55-56 computes through the one-hot GEMM."""
    dim = x.shape[-1]
    flat = x.reshape(-1, dim)
    ind = idx.reshape(-1)
    counts = torch.bincount(ind, minlength=n_embed).to(flat.dtype)
    sums = torch.zeros(n_embed, dim, dtype=flat.dtype).index_add_(0, ind, flat).t().contiguous()
    return counts, sums


def quantize_margin(x, embed):
    """fp64 best-vs-second-best distance gap per vector (for near-tie accounting)."""
    flat = x.reshape(-1, embed.shape[0]).double()
    e = embed.double()
    d = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ e + e.pow(2).sum(0, keepdim=True)
    two = torch.topk(-d, 2, dim=1).values
    return (two[:, 0] - two[:, 1]), d.argmin(1)


def quantize_margin_chunked(x, embed, rows=8192):
    """quantize_margin in row blocks: the [M,K] fp64 distance matrix of a full-size case (M = 131,072,
    K = 8,192 -> 8.6 GB) never exists; same results."""
    flat = x.reshape(-1, embed.shape[0])
    margins, idxs = [], []
    for r0 in range(0, flat.shape[0], rows):
        mg, ix = quantize_margin(flat[r0:r0 + rows], embed)
        margins.append(mg)
        idxs.append(ix)
    return torch.cat(margins), torch.cat(idxs)


# --------------------------------------------------------------------------
# conv stacks (vqvae.py:81-166)
# --------------------------------------------------------------------------
def resblock(st, p, x):
    """vqvae.py:85-94: out = x + conv1x1(relu(conv3x3(relu(x)))); first ReLU is NOT in place."""
    h = F.conv2d(F.relu(x), st[f"{p}.conv.1.weight"], st[f"{p}.conv.1.bias"], padding=1)
    h = F.conv2d(F.relu(h), st[f"{p}.conv.3.weight"], st[f"{p}.conv.3.bias"])
    return h + x


def encoder(st, p, x, n_res_block, stride):
    if stride == 4:  # vqvae.py:103-110
        x = F.relu(F.conv2d(x, st[f"{p}.blocks.0.weight"], st[f"{p}.blocks.0.bias"], stride=2, padding=1))
        x = F.relu(F.conv2d(x, st[f"{p}.blocks.2.weight"], st[f"{p}.blocks.2.bias"], stride=2, padding=1))
        x = F.conv2d(x, st[f"{p}.blocks.4.weight"], st[f"{p}.blocks.4.bias"], padding=1)
        base = 5
    else:  # vqvae.py:112-117
        x = F.relu(F.conv2d(x, st[f"{p}.blocks.0.weight"], st[f"{p}.blocks.0.bias"], stride=2, padding=1))
        x = F.conv2d(x, st[f"{p}.blocks.2.weight"], st[f"{p}.blocks.2.bias"], padding=1)
        base = 3
    for i in range(n_res_block):
        x = resblock(st, f"{p}.blocks.{base + i}", x)
    return F.relu(x)  # vqvae.py:122


def decoder(st, p, x, n_res_block, stride):
    x = F.conv2d(x, st[f"{p}.blocks.0.weight"], st[f"{p}.blocks.0.bias"], padding=1)  # :137
    for i in range(n_res_block):
        x = resblock(st, f"{p}.blocks.{1 + i}", x)
    x = F.relu(x)  # :144
    t = n_res_block + 2
    x = F.conv_transpose2d(x, st[f"{p}.blocks.{t}.weight"], st[f"{p}.blocks.{t}.bias"], stride=2, padding=1)
    if stride == 4:  # :147-156
        x = F.relu(x)
        x = F.conv_transpose2d(x, st[f"{p}.blocks.{t + 2}.weight"], st[f"{p}.blocks.{t + 2}.bias"],
                               stride=2, padding=1)
    return x


# --------------------------------------------------------------------------
# VQVAE wiring (vqvae.py:216-259)
# --------------------------------------------------------------------------
def _vq(st, name, x_nhwc, training, cfg, all_reduce):
    return quantize_forward(x_nhwc, st[f"{name}.embed"], st[f"{name}.cluster_size"],
                            st[f"{name}.embed_avg"], training, 0.99, cfg.eps, all_reduce)


def vqvae_encode(st, cfg, x, training=True, all_reduce=None):
    n = cfg.n_res_block
    enc_b = encoder(st, "enc_b", x, n, 4)
    enc_t = encoder(st, "enc_t", enc_b, n, 2)
    q_t = F.conv2d(enc_t, st["quantize_conv_t.weight"], st["quantize_conv_t.bias"]).permute(0, 2, 3, 1)
    q_t, diff_t, id_t = _vq(st, "quantize_t", q_t, training, cfg, all_reduce)
    q_t = q_t.permute(0, 3, 1, 2)
    dec_t = decoder(st, "dec_t", q_t, n, 2)
    cat = torch.cat([dec_t, enc_b], 1)
    q_b = F.conv2d(cat, st["quantize_conv_b.weight"], st["quantize_conv_b.bias"]).permute(0, 2, 3, 1)
    q_b, diff_b, id_b = _vq(st, "quantize_b", q_b, training, cfg, all_reduce)
    q_b = q_b.permute(0, 3, 1, 2)
    return q_t, q_b, diff_t.unsqueeze(0) + diff_b.unsqueeze(0), id_t, id_b


def vqvae_decode(st, cfg, quant):
    return decoder(st, "dec", quant, cfg.n_res_block, 4)


def vqvae_forward(st, cfg, x, training=True, all_reduce=None):
    q_t, q_b, diff, id_t, id_b = vqvae_encode(st, cfg, x, training, all_reduce)
    up = F.conv_transpose2d(q_t, st["upsample_t.weight"], st["upsample_t.bias"], stride=2, padding=1)
    dec = vqvae_decode(st, cfg, torch.cat([up, q_b], 1))
    return dec, diff, id_t, id_b


def vqvae_decode_code(st, cfg, code_t, code_b):
    """Upstream semantics of vqvae.py:251-259 (the fork passes two args to decode())."""
    q_t = F.embedding(code_t, st["quantize_t.embed"].t()).permute(0, 3, 1, 2)
    q_b = F.embedding(code_b, st["quantize_b.embed"].t()).permute(0, 3, 1, 2)
    up = F.conv_transpose2d(q_t, st["upsample_t.weight"], st["upsample_t.bias"], stride=2, padding=1)
    return vqvae_decode(st, cfg, torch.cat([up, q_b], 1))


# config 1: "single-level VQ-VAE" = composition of reference parts
def single_level_spec(cfg: VQVAEConfig):
    c, e = cfg.channel, cfg.embed_dim
    spec = _encoder_spec("enc", cfg.in_channel, c, cfg.n_res_block, cfg.n_res_channel, 4)
    spec += [("quantize_conv", "conv", (e, c, 1, 1)), ("quantize", "vq", (e, cfg.n_embed))]
    spec += _decoder_spec("dec", e, cfg.in_channel, c, cfg.n_res_block, cfg.n_res_channel, 4)
    return spec


def make_single_level_state(cfg, seed=1234):
    st = {}
    for name, kind, shape in single_level_spec(cfg):
        if kind == "vq":
            emb = rng.normal(seed, "sl." + name + ".embed", shape)
            st[f"{name}.embed"] = torch.from_numpy(emb.copy())
            st[f"{name}.cluster_size"] = torch.zeros(shape[1])
            st[f"{name}.embed_avg"] = torch.from_numpy(emb.copy())
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            nb = shape[0] if kind == "conv" else shape[1]
            b = 1.0 / math.sqrt(fan_in)
            st[f"{name}.weight"] = torch.from_numpy(rng.uniform(seed, "sl." + name + ".weight", shape, -b, b))
            st[f"{name}.bias"] = torch.from_numpy(rng.uniform(seed, "sl." + name + ".bias", (nb,), -b, b))
    return st


def single_level_forward(st, cfg, x, training=True, all_reduce=None):
    h = encoder(st, "enc", x, cfg.n_res_block, 4)
    q = F.conv2d(h, st["quantize_conv.weight"], st["quantize_conv.bias"]).permute(0, 2, 3, 1)
    q, diff, idx = quantize_forward(q, st["quantize.embed"], st["quantize.cluster_size"],
                                    st["quantize.embed_avg"], training, 0.99, cfg.eps, all_reduce)
    dec = decoder(st, "dec", q.permute(0, 3, 1, 2), cfg.n_res_block, 4)
    return dec, diff.unsqueeze(0), idx


# --------------------------------------------------------------------------
# stage-1 step (train_vqvae.py:83-91, 185)
# --------------------------------------------------------------------------
LATENT_LOSS_WEIGHT = 0.25  # train_vqvae.py:34


def stage1_loss(dec, diff, img):
    recon = F.mse_loss(dec, img)
    latent = diff.mean()
    return recon + LATENT_LOSS_WEIGHT * latent, recon, latent


class AdamState:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) restated."""

    def __init__(self, params):
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, params, grads, lr=3e-4, b1=0.9, b2=0.999, eps=1e-8):
        self.t += 1
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        step_size = lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        with torch.no_grad():
            for k, p in params.items():
                g = grads.get(k)
                if g is None:
                    continue
                self.m[k].lerp_(g, 1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / bc2_sqrt).add_(eps)
                p.addcdiv_(self.m[k], denom, value=-step_size)


def train_step(st, cfg, img, adam: AdamState, lr=3e-4, forward=vqvae_forward, all_reduce=None,
               grad_all_reduce=None):
    """One canonical stage-1 step; returns dict of scalars + grads. Mutates st/adam in place.

    `grad_all_reduce(g)` averages a gradient across ranks in place (DDP semantics,
    train_vqvae.py:166-171); None = world 1.
    """
    params = {k: v for k, v in st.items() if not is_buffer(k)}
    leaves = {k: v.detach().requires_grad_(True) for k, v in params.items()}
    work = dict(st)
    work.update(leaves)
    out = forward(work, cfg, img, True, all_reduce)
    dec, diff = out[0], out[1]
    loss, recon, latent = stage1_loss(dec, diff, img)
    used = {k: v for k, v in leaves.items() if not k.startswith("dec_ir.")}
    gl = torch.autograd.grad(loss, list(used.values()), allow_unused=True)
    grads = {k: g for k, g in zip(used.keys(), gl) if g is not None}
    if grad_all_reduce is not None:
        for g in grads.values():
            grad_all_reduce(g)
    adam.step(params, grads, lr=lr)
    return {"loss": loss.detach(), "recon": recon.detach(), "latent": latent.detach(),
            "dec": dec.detach(), "diff": diff.detach(), "ids": [o.detach() for o in out[2:]],
            "grads": grads}


# --------------------------------------------------------------------------
# CycleScheduler (scheduler.py:221-320) -- the only scheduler stage 1 uses
# --------------------------------------------------------------------------
class CycleSchedule:
    """lr(t) trajectory of scheduler.py:CycleScheduler with momentum=None
    (train_vqvae.py:189-195: lr_max=args.lr, n_iter=len(loader)*epoch, warmup 0.05)."""

    def __init__(self, lr_max, n_iter, divider=25, warmup_proportion=0.3):
        self.p1 = int(n_iter * warmup_proportion)
        self.p2 = n_iter - self.p1
        self.lr_max = lr_max
        self.lr_min = lr_max / divider
        self.phase = 0
        self.n = 0

    def step(self):
        self.n += 1
        if self.phase == 0:
            lr = self.lr_min + (self.n / self.p1) * (self.lr_max - self.lr_min)
            done = self.n >= self.p1
        else:
            end = self.lr_min / 1e4
            lr = end + (self.lr_max - end) / 2 * (math.cos(math.pi * self.n / self.p2) + 1)
            done = self.n >= self.p2
        if done:
            self.phase += 1
            self.n = 0
            if self.phase >= 2:
                self.phase = 0
        return lr


# --------------------------------------------------------------------------
# misc helpers for fixtures
# --------------------------------------------------------------------------
def tensor_digest(t):
    """crc32 of the raw little-endian bytes (for integer tensors / exact checks)."""
    return zlib.crc32(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
