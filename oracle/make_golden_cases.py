"""Case tables + seeded input builders shared by oracle/make_golden.py (which runs the
reference) and tests/ (which never see the reference).  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch

from . import rng
from . import vqvae_oracle as O

SEED = 1234

CONV_FLAVOURS = [
    # tag, kind, weight shape, stride, pad, input HxW   (shapes of SURVEY 8c item 2)
    ("c4s2_3_64", "conv", (64, 3, 4, 4), 2, 1, 16),
    ("c4s2_64_128", "conv", (128, 64, 4, 4), 2, 1, 8),
    ("c4s2_128_64", "conv", (64, 128, 4, 4), 2, 1, 8),
    ("c3_128_128", "conv", (128, 128, 3, 3), 1, 1, 8),
    ("c3_64_128", "conv", (128, 64, 3, 3), 1, 1, 8),
    ("c3_128_32", "conv", (32, 128, 3, 3), 1, 1, 8),
    ("c1_32_128", "conv", (128, 32, 1, 1), 1, 0, 8),
    ("c1_128_64", "conv", (64, 128, 1, 1), 1, 0, 8),
    ("c1_192_64", "conv", (64, 192, 1, 1), 1, 0, 8),
    ("t4s2_128_64", "convT", (128, 64, 4, 4), 2, 1, 8),
    ("t4s2_64_64", "convT", (64, 64, 4, 4), 2, 1, 8),
    ("t4s2_64_3", "convT", (64, 3, 4, 4), 2, 1, 8),
    ("c3_48_8_odd", "conv", (8, 48, 3, 3), 1, 1, 7),      # ragged spatial, small channels
    ("t4s2_16_3_odd", "convT", (16, 4, 4, 4), 2, 1, 5),
]

# tag, kind, ctor args (reference signature order), input shape
BLOCK_CASES = [
    ("rb_128_32", "resblock", (128, 32), (2, 128, 8, 8)),
    ("rb_32_8", "resblock", (32, 8), (2, 32, 6, 5)),
    ("enc4", "encoder", (3, 32, 1, 8, 4), (2, 3, 32, 32)),
    ("enc2", "encoder", (32, 32, 2, 8, 2), (2, 32, 8, 8)),
    ("enc4_nores", "encoder", (3, 32, 0, 8, 4), (2, 3, 16, 16)),
    ("dec2", "decoder", (16, 16, 32, 1, 8, 2), (2, 16, 4, 4)),
    ("dec4", "decoder", (32, 3, 32, 2, 8, 4), (2, 32, 8, 8)),
]


def conv_inputs(tag, kind, ws, hw):
    cin = ws[1] if kind == "conv" else ws[0]
    cout = ws[0] if kind == "conv" else ws[1]
    x = rng.normal(SEED, f"{tag}.x", (2, cin, hw, hw))
    fan_in = ws[1] * ws[2] * ws[3]
    w = (rng.uniform(SEED, f"{tag}.w", ws, -1, 1) / np.sqrt(fan_in)).astype(np.float32)
    b = rng.uniform(SEED, f"{tag}.b", (cout,), -1, 1)
    return x, w, b


def quantize_inputs(tag, D, K, xshape, tie=False):
    embed = rng.normal(SEED, f"{tag}.embed", (D, K))
    if tie:  # duplicate codebook columns -> the first index must win (vqvae.py:49)
        embed[:, 300] = embed[:, 5]
        embed[:, 7] = embed[:, 5]
        embed[:, 100] = embed[:, 64]
    x = rng.normal(SEED, f"{tag}.x", xshape)
    if tie:
        flat = x.reshape(-1, D)
        flat[0] = embed[:, 5]
        flat[1] = embed[:, 64]
        flat[2] = embed[:, 300]
        x = flat.reshape(xshape)
    gw = rng.normal(SEED, f"{tag}.gw", xshape)
    cs0 = (np.abs(rng.normal(SEED, f"{tag}.cs", (K,))) * 3.0).astype(np.float32)
    return x, embed, cs0, gw


def block_keys(kind, args):
    """[(state_dict key relative to the block, shape)] in the reference's order."""
    if kind == "resblock":
        cin, ch = args
        return [("conv.1.weight", (ch, cin, 3, 3)), ("conv.1.bias", (ch,)),
                ("conv.3.weight", (cin, ch, 1, 1)), ("conv.3.bias", (cin,))]
    if kind == "encoder":
        spec = O._encoder_spec("m", *args)
    else:
        spec = O._decoder_spec("m", *args)
    out = []
    for name, k, shape in spec:
        out.append((name[2:] + ".weight", shape))
        out.append((name[2:] + ".bias", (shape[0],) if k == "conv" else (shape[1],)))
    return out


def block_state(tag, kind, args):
    st = {}
    for k, shape in block_keys(kind, args):
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            a = rng.uniform(SEED, f"{tag}.{k}", shape, -1, 1) / np.sqrt(fan_in)
        else:
            a = rng.uniform(SEED, f"{tag}.{k}", shape, -0.1, 0.1)
        st[k] = torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))
    return st


# CycleScheduler trajectories captured from the reference's scheduler.py (tag, ctor kwargs, number of steps);
# "stage1" is exactly what train_vqvae.py:188-195 builds
SCHED_CASES = [
    ("stage1", dict(lr_max=3e-4, n_iter=450, momentum=None, warmup_proportion=0.05), 1000),
    ("defaults", dict(lr_max=1e-3, n_iter=100), 250),
    ("cos_linear", dict(lr_max=1e-2, n_iter=37, momentum=(0.9, 0.8), divider=10, warmup_proportion=0.5,
                        phase=("cos", "linear")), 120),
]


# VQVAE_Deep: extra conv flavours its decoder needs, an embed_dim-256 Quantize case, AdaIN shapes
DEEP_CONV_FLAVOURS = [
    ("t4s2_3_3", "convT", (3, 3, 4, 4), 2, 1, 8),          # up2(out_channel): ConvTranspose2d(3, 3)  (vqvae_deep.py:212)
    ("c3_512_256", "conv", (256, 512, 3, 3), 1, 1, 4),     # dec.conv1 of the default VQVAE_Deep
    ("c1_2048_512", "conv", (512, 2048, 1, 1), 1, 0, 1),   # AdaIN's nn.Linear(2048, 512) as a 1x1 conv on a 1x1 image
]
DEEP_ADAIN_CASES = [
    # tag, style_dim, channels, (N, H, W)
    ("adain_256", 2048, 256, (2, 8, 8)),
    ("adain_16_odd", 24, 16, (3, 5, 7)),
]
DEEP_SEED = 4321
DEEP_EMBED_SCALE, DEEP_GAIN = 0.3, 2.0     # see vqvae_deep_oracle.make_deep_state


def thin(a, limit=65536):
    """Big gradient tensors are stored as every k-th element of the flattened array (fixture size)."""
    flat = np.asarray(a).reshape(-1)
    k = max(1, flat.size // limit)
    return flat[::k].copy() if k > 1 else np.asarray(a)

# Encoder / Decoder geometries of vqvae_deep.py that VQVAE_Deep itself does not use (strides 8 and 4; a styled decoder
# at stride 4): tag, kind, ctor args in the reference's order, input shape
DEEP_BLOCK_CASES = [
    ("denc8", "encoder", (3, 16, 1, 8, 8), (2, 3, 32, 32)),
    ("denc4", "encoder", (4, 16, 0, 8, 4), (1, 4, 16, 16)),
    ("ddec8", "decoder", (16, 3, 16, -1, 1, 8, 8), (2, 16, 2, 2)),
    ("ddec4s", "decoder", (8, 4, 16, 12, 2, 8, 4), (2, 8, 4, 4)),      # AdaIN blocks, style_dim 12
]
