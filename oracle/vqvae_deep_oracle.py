"""CPU oracle for the VQVAE_Deep variant: a functional PyTorch-CPU restatement of
/root/reference/vqvae_deep.py.  TEST INFRASTRUCTURE ONLY (same rules as vqvae_oracle.py: imported by tests/,
smoke() and oracle/make_golden.py, never by the product package).

Pinned: oracle/make_golden.py gen_deep() runs the reference's own vqvae_deep.py on the tiny configuration below
and tests/test_oracle_golden.py checks this restatement against those captured outputs.

  adain            <- vqvae_deep.py:99-109   (InstanceNorm2d(affine=False) + Linear + (1+gamma)*norm + beta)
  adain_resblk     <- vqvae_deep.py:127-134
  deep_encoder     <- vqvae_deep.py:136-174
  deep_decoder     <- vqvae_deep.py:177-229
  deep_encode / deep_quantize / deep_forward <- vqvae_deep.py:275-301 (+306-307 decode with a style)
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import rng
from .vqvae_oracle import quantize_forward, resblock


@dataclass(frozen=True)
class DeepConfig:
    in_channel: int = 3
    channel: int = 256
    n_res_block: int = 6
    n_res_channel: int = 128
    embed_dim: int = 256
    n_embed: int = 512
    out_channel: int = 3
    style_dim: int = 2048
    eps: float = 1e-5


DEEP_DEFAULT = DeepConfig()
DEEP_TINY = DeepConfig(channel=32, n_res_block=1, n_res_channel=16, embed_dim=16, n_embed=64, style_dim=24)


# ---------------------------------------------------------------- state_dict layout (registration order of the reference)
def _down4(p, base, in_ch, ch):
    return [(f"{p}.blocks.{base}", "conv", (ch // 2, in_ch, 4, 4)), (f"{p}.blocks.{base + 2}", "conv", (ch, ch // 2, 4, 4)),
            (f"{p}.blocks.{base + 4}", "conv", (ch, ch, 3, 3))], base + 5


def _down2(p, base, in_ch, ch):
    return [(f"{p}.blocks.{base}", "conv", (ch // 2, in_ch, 4, 4)), (f"{p}.blocks.{base + 2}", "conv", (ch, ch // 2, 3, 3))], base + 3


def _enc_spec(p, in_ch, ch, n_res, n_res_ch, stride):
    spec, base = [], 0
    if stride == 6:
        s, base = _down2(p, base, in_ch, ch)
        spec += s
        s, base = _down4(p, base, ch, ch)
        spec += s
    elif stride == 2:
        s, base = _down2(p, base, in_ch, ch)
        spec += s
    else:
        raise ValueError("oracle covers the strides VQVAE_Deep uses (6 and 2)")
    for i in range(n_res):
        spec += [(f"{p}.blocks.{base + i}.conv.1", "conv", (n_res_ch, ch, 3, 3)),
                 (f"{p}.blocks.{base + i}.conv.3", "conv", (ch, n_res_ch, 1, 1))]
    return spec


def _dec_spec(p, in_ch, out_ch, ch, style_dim, n_res, n_res_ch, stride):
    spec = [(f"{p}.conv1", "conv", (ch, in_ch, 3, 3))]
    if stride == 6:      # up4(channel) + up2(out_channel)   (vqvae_deep.py:210-212)
        spec += [(f"{p}.up_sample.0", "convT", (ch, ch // 2, 4, 4)), (f"{p}.up_sample.2", "convT", (ch // 2, out_ch, 4, 4)),
                 (f"{p}.up_sample.3", "convT", (out_ch, out_ch, 4, 4))]
    elif stride == 2:
        spec += [(f"{p}.up_sample.0", "convT", (ch, out_ch, 4, 4))]
    else:
        raise ValueError("oracle covers the strides VQVAE_Deep uses (6 and 2)")
    for i in range(n_res):
        b = f"{p}.blocks.{i}"
        if style_dim <= 0:
            spec += [(f"{b}.conv.1", "conv", (n_res_ch, ch, 3, 3)), (f"{b}.conv.3", "conv", (ch, n_res_ch, 1, 1))]
        else:
            spec += [(f"{b}.conv1", "conv", (n_res_ch, ch, 3, 3)), (f"{b}.conv2", "conv", (ch, n_res_ch, 1, 1)),
                     (f"{b}.norm1.fc", "linear", (2 * ch, style_dim)), (f"{b}.norm2.fc", "linear", (2 * n_res_ch, style_dim)),
                     (f"{b}.conv.1", "conv", (n_res_ch, ch, 3, 3)), (f"{b}.conv.3", "conv", (ch, n_res_ch, 1, 1))]   # dead
    return spec


def deep_layer_spec(cfg: DeepConfig):
    c, e = cfg.channel, cfg.embed_dim
    spec = _enc_spec("enc_b", cfg.in_channel, c, cfg.n_res_block, cfg.n_res_channel, 6)
    spec += _enc_spec("enc_t", c, c, cfg.n_res_block, cfg.n_res_channel, 2)
    spec += [("quantize_conv_t", "conv", (e, c, 1, 1)), ("quantize_t", "vq", (e, cfg.n_embed))]
    spec += _dec_spec("dec_t", e, e, c, -1, cfg.n_res_block, cfg.n_res_channel, 2)
    spec += [("quantize_conv_b", "conv", (e, e + c, 1, 1)), ("quantize_b", "vq", (e, cfg.n_embed))]
    spec += [("upsample_t.0", "convT", (e, e, 4, 4))]
    spec += _dec_spec("dec", e + e, cfg.out_channel, c, cfg.style_dim, cfg.n_res_block, cfg.n_res_channel, 6)
    return spec


def deep_state_spec(cfg: DeepConfig):
    out = {}
    for name, kind, shape in deep_layer_spec(cfg):
        if kind == "vq":
            out[f"{name}.embed"] = shape
            out[f"{name}.cluster_size"] = (shape[1],)
            out[f"{name}.embed_avg"] = shape
        else:
            out[f"{name}.weight"] = shape
            out[f"{name}.bias"] = (shape[1],) if kind == "convT" else (shape[0],)
    return out


def make_deep_state(cfg: DeepConfig, seed=1234, embed_scale=1.0, gain=1.0):
    """Synthetic state_dict from the counter RNG: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights and biases,
    embed_scale * N(0,1) codebooks (vqvae_deep.py:37-40 draws N(0,1); a smaller scale puts the codes among the
    latents of an untrained encoder and a conv-weight `gain` > 1 spreads those latents, so that a test exercises
    many different indices instead of one collapsed code)."""
    st = {}
    for name, kind, shape in deep_layer_spec(cfg):
        if kind == "vq":
            emb = (rng.normal(seed, name + ".embed", shape) * embed_scale).astype("float32")
            st[f"{name}.embed"] = torch.from_numpy(emb.copy())
            st[f"{name}.cluster_size"] = torch.zeros(shape[1])
            st[f"{name}.embed_avg"] = torch.from_numpy(emb.copy())
            continue
        fan_in = shape[1] if kind == "linear" else shape[1] * shape[2] * shape[3]
        nb = shape[1] if kind == "convT" else shape[0]
        b = 1.0 / math.sqrt(fan_in)
        g = 1.0 if kind == "linear" else gain
        st[f"{name}.weight"] = torch.from_numpy(rng.uniform(seed, name + ".weight", shape, -b * g, b * g))
        st[f"{name}.bias"] = torch.from_numpy(rng.uniform(seed, name + ".bias", (nb,), -b, b))
    return st


def make_style(batch, cfg: DeepConfig, seed=1234):
    return torch.from_numpy(rng.normal(seed, "style", (batch, cfg.style_dim)))


def is_dead_key(key):
    """AdainResBlk.conv (vqvae_deep.py:120-125) is registered but never executed."""
    return ".blocks." in key and key.startswith("dec.") and ".conv." in key


# ---------------------------------------------------------------- layers
def adain(st, p, x, s, eps=1e-5):
    h = F.linear(s, st[f"{p}.fc.weight"], st[f"{p}.fc.bias"])
    h = h.view(h.size(0), h.size(1), 1, 1)
    gamma, beta = torch.chunk(h, chunks=2, dim=1)
    return (1 + gamma) * F.instance_norm(x, eps=eps) + beta


def adain_resblk(st, p, x, s):
    out = adain(st, f"{p}.norm1", x, s)
    out = F.conv2d(F.relu(out), st[f"{p}.conv1.weight"], st[f"{p}.conv1.bias"], padding=1)
    out = adain(st, f"{p}.norm2", out, s)
    out = F.conv2d(F.relu(out), st[f"{p}.conv2.weight"], st[f"{p}.conv2.bias"])
    return out + x


def _conv(st, key, x, stride=1, padding=0):
    return F.conv2d(x, st[f"{key}.weight"], st[f"{key}.bias"], stride=stride, padding=padding)


def _convT(st, key, x):
    return F.conv_transpose2d(x, st[f"{key}.weight"], st[f"{key}.bias"], stride=2, padding=1)


def deep_encoder(st, p, x, n_res_block, stride):
    base = 0
    if stride == 6:    # down2 then down4, NO ReLU between them (vqvae_deep.py:160-162)
        x = F.relu(_conv(st, f"{p}.blocks.0", x, 2, 1))
        x = _conv(st, f"{p}.blocks.2", x, 1, 1)
        x = F.relu(_conv(st, f"{p}.blocks.3", x, 2, 1))
        x = F.relu(_conv(st, f"{p}.blocks.5", x, 2, 1))
        x = _conv(st, f"{p}.blocks.7", x, 1, 1)
        base = 8
    else:
        x = F.relu(_conv(st, f"{p}.blocks.0", x, 2, 1))
        x = _conv(st, f"{p}.blocks.2", x, 1, 1)
        base = 3
    for i in range(n_res_block):
        x = resblock(st, f"{p}.blocks.{base + i}", x)
    return F.relu(x)


def deep_decoder(st, p, x, n_res_block, stride, s=None):
    x = _conv(st, f"{p}.conv1", x, 1, 1)
    for i in range(n_res_block):
        x = resblock(st, f"{p}.blocks.{i}", x) if s is None else adain_resblk(st, f"{p}.blocks.{i}", x, s)
    x = F.relu(x)
    x = _convT(st, f"{p}.up_sample.0", x)
    if stride == 6:
        x = _convT(st, f"{p}.up_sample.2", F.relu(x))
        x = _convT(st, f"{p}.up_sample.3", x)          # ConvTranspose2d(out_channel, out_channel): no ReLU before it
    return x


def _vq(st, name, x, training, cfg):
    return quantize_forward(x, st[f"{name}.embed"], st[f"{name}.cluster_size"], st[f"{name}.embed_avg"], training,
                            0.99, cfg.eps, None)


def deep_encode(st, cfg, x):
    enc_b = deep_encoder(st, "enc_b", x, cfg.n_res_block, 6)
    enc_t = deep_encoder(st, "enc_t", enc_b, cfg.n_res_block, 2)
    return enc_b, enc_t


def deep_quantize(st, cfg, enc_b, enc_t, training=True):
    q_t = _conv(st, "quantize_conv_t", enc_t).permute(0, 2, 3, 1)
    q_t, diff_t, id_t = _vq(st, "quantize_t", q_t, training, cfg)
    q_t = q_t.permute(0, 3, 1, 2)
    dec_t = deep_decoder(st, "dec_t", q_t, cfg.n_res_block, 2)
    cat = torch.cat([dec_t, enc_b], 1)
    q_b = _conv(st, "quantize_conv_b", cat).permute(0, 2, 3, 1)
    q_b, diff_b, id_b = _vq(st, "quantize_b", q_b, training, cfg)
    q_b = q_b.permute(0, 3, 1, 2)
    return q_t, q_b, diff_t.unsqueeze(0) + diff_b.unsqueeze(0), id_t, id_b


def deep_forward(st, cfg, x, style, training=True):
    """encode -> quantize -> upsample_t -> cat -> decode(quant, style): what vqvae_deep.py:275-279 intends."""
    enc_b, enc_t = deep_encode(st, cfg, x)
    q_t, q_b, diff, id_t, id_b = deep_quantize(st, cfg, enc_b, enc_t, training)
    up = _convT(st, "upsample_t.0", q_t)
    quant = torch.cat([up, q_b], 1)
    dec = deep_decoder(st, "dec", quant, cfg.n_res_block, 6, style)
    return dec, diff, quant, id_t, id_b
