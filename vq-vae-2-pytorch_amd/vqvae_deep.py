"""Host-side mirror of the reference's vqvae_deep.py: same class names, constructor signatures,
method contracts and state_dict keys/shapes, on libvq2's HIP kernels.

  AdaIN         <- /root/reference/vqvae_deep.py:99-109
  AdainResBlk   <- vqvae_deep.py:112-134
  Encoder       <- vqvae_deep.py:136-174   (strides 2, 4, 6, 8)
  Decoder       <- vqvae_deep.py:177-229   (ResBlock or AdaIN-ResBlock body; strides 2, 4, 6, 8)
  VQVAE_Deep    <- vqvae_deep.py:234-320

Quantize and ResBlock are byte-identical to vqvae.py's in the reference (vqvae_deep.py:28-96) and are the same
objects here.  embed_dim 256 runs on the D = 256 instantiation of the fused Quantize kernel; ResBlock(256, 128)
composes two conv launches (the one-launch kernel is built for 128/32); InstanceNorm + style modulation are the
vq2_instnorm_stats / vq2_adain_* kernels; the style projection nn.Linear is a 1x1 conv on a [N,1,1,style_dim] tensor.

Fork quirks kept: `VQVAE_Deep.forward(input)` calls `decode(quant)` without the style argument
(vqvae_deep.py:277 vs 306) and so raises TypeError in the reference -- here too, unless the caller passes
`style=`; AdainResBlk carries an unused `conv` Sequential (vqvae_deep.py:120-125) that lives in the state_dict.
"""
import math

import torch
from torch import nn

from . import ops
from .ops import ConvSpec
from .vqvae import Conv2d, ConvTranspose2d, Quantize, ReLU, ResBlock, _run_blocks


class Linear(nn.Module):
    """nn.Linear(in_features, out_features) (same parameters and default init) computed as a 1x1 conv."""

    def __init__(self, in_features, out_features):
        super().__init__()
        if in_features % 4 or out_features % 4:
            raise NotImplementedError("vqvae2_amd.Linear: feature counts must be multiples of 4")
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)
        self.spec = ConvSpec(False, in_features, out_features, 1, 1, 0)

    def forward(self, input):
        lead = input.shape[:-1]
        x = input.reshape(-1, 1, 1, self.in_features)
        # the Parameter itself (not a 4-D view of it): the kernels take raw pointers, and the weight gradient can then
        # land directly in the parameter's slot of a ParamArena (a view has no slot and autograd would copy)
        y = ops.conv_op(x, self.weight, self.bias, self.spec)
        return y.reshape(*lead, self.out_features)


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(num_features, affine=False): no parameters, no buffers; holds eps for AdaIN."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps


class AdaIN(nn.Module):
    def __init__(self, style_dim, num_features):
        super().__init__()
        self.norm = InstanceNorm2d(num_features)
        self.fc = Linear(style_dim, num_features * 2)

    def nhwc(self, x, s, relu=False):
        return ops.AdaINFn.apply(x, self.fc(s), relu, self.norm.eps)

    def forward(self, x, s):
        return ops.from_nhwc(self.nhwc(ops.to_nhwc(x), s), x.shape[1])


class AdainResBlk(nn.Module):
    def __init__(self, in_channel, channel, style_dim):
        super().__init__()
        self.conv1 = Conv2d(in_channel, channel, 3, padding=1)
        self.conv2 = Conv2d(channel, in_channel, 1)
        self.norm1 = AdaIN(style_dim, in_channel)
        self.norm2 = AdaIN(style_dim, channel)
        # never executed by the reference either (vqvae_deep.py:120-125); kept for state_dict parity
        self.conv = nn.Sequential(ReLU(), Conv2d(in_channel, channel, 3, padding=1), ReLU(inplace=True),
                                  Conv2d(channel, in_channel, 1))

    def nhwc(self, x, s):
        out = self.norm1.nhwc(x, s, relu=True)            # vqvae_deep.py:128-129 (F.relu_ fused)
        out = self.conv1.nhwc(out)
        out = self.norm2.nhwc(out, s, relu=True)          # :130-131
        return self.conv2.nhwc(out, residual=x)           # :131-132 (out += input)

    def forward(self, input, s):
        return ops.from_nhwc(self.nhwc(ops.to_nhwc(input), s), input.shape[1])


def _down4(in_channel, channel):
    return [Conv2d(in_channel, channel // 2, 4, stride=2, padding=1), ReLU(inplace=True),
            Conv2d(channel // 2, channel, 4, stride=2, padding=1), ReLU(inplace=True),
            Conv2d(channel, channel, 3, padding=1)]


def _down2(in_channel, channel):
    return [Conv2d(in_channel, channel // 2, 4, stride=2, padding=1), ReLU(inplace=True),
            Conv2d(channel // 2, channel, 3, padding=1)]


class Encoder(nn.Module):
    def __init__(self, in_channel, channel, n_res_block, n_res_channel, stride):
        super().__init__()
        blocks = []
        if stride == 8:
            blocks += _down4(in_channel, channel) + _down4(channel, channel)
        if stride == 6:
            blocks += _down2(in_channel, channel) + _down4(channel, channel)
        elif stride == 4:
            blocks += _down4(in_channel, channel)
        elif stride == 2:
            blocks += _down2(in_channel, channel)
        for _ in range(n_res_block):
            blocks.append(ResBlock(channel, n_res_channel))
        blocks.append(ReLU(inplace=True))
        self.blocks = nn.Sequential(*blocks)
        self.out_channels = channel

    def nhwc(self, x):
        return _run_blocks(self.blocks, x)

    def forward(self, input):
        return ops.from_nhwc(self.nhwc(ops.to_nhwc(input)), self.out_channels)


class Decoder(nn.Module):
    def __init__(self, in_channel, out_channel, channel, style_dim, n_res_block, n_res_channel, stride):
        super().__init__()
        self.style = style_dim > 1
        blocks = []
        self.conv1 = Conv2d(in_channel, channel, 3, padding=1)
        for _ in range(n_res_block):
            blocks.append(ResBlock(channel, n_res_channel) if style_dim <= 0 else
                          AdainResBlk(channel, n_res_channel, style_dim))
        self.relu = ReLU(inplace=True)

        def up4(ch):
            return [ConvTranspose2d(ch, ch // 2, 4, stride=2, padding=1), ReLU(inplace=True),
                    ConvTranspose2d(ch // 2, out_channel, 4, stride=2, padding=1)]

        def up2(ch):
            return [ConvTranspose2d(ch, out_channel, 4, stride=2, padding=1)]

        up_sample = []
        if stride == 8:
            up_sample += up4(channel) + up4(out_channel)
        elif stride == 6:
            up_sample += up4(channel) + up2(out_channel)
        elif stride == 4:
            up_sample += up4(channel)
        elif stride == 2:
            up_sample += up2(channel)
        self.up_sample = nn.Sequential(*up_sample)
        self.blocks = nn.Sequential(*blocks)
        self.out_channels = out_channel

    def nhwc(self, x, s=None):
        out = self.conv1.nhwc(x)
        if not self.style:
            out = _run_blocks(self.blocks, out) if len(self.blocks) else out
        else:
            if s is None:
                raise TypeError("Decoder with AdaIN blocks needs a style tensor")
            for blk in self.blocks:
                out = blk.nhwc(out, s)
        return _run_blocks([self.relu] + list(self.up_sample), out)     # the ReLU fuses into the first up-conv

    def forward(self, input, s=None):
        return ops.from_nhwc(self.nhwc(ops.to_nhwc(input), s), self.out_channels)


class VQVAE_Deep(nn.Module):
    def __init__(self, in_channel=3, channel=256, n_res_block=6, n_res_channel=128, embed_dim=256, n_embed=512,
                 decay=0.99, out_channel=3, style_dim=2048):
        super().__init__()
        self.enc_b = Encoder(in_channel, channel, n_res_block, n_res_channel, stride=6)
        self.enc_t = Encoder(channel, channel, n_res_block, n_res_channel, stride=2)
        self.quantize_conv_t = Conv2d(channel, embed_dim, 1)
        self.quantize_t = Quantize(embed_dim, n_embed)       # `decay` is not forwarded (vqvae_deep.py:252)
        self.dec_t = Decoder(embed_dim, embed_dim, channel, -1, n_res_block, n_res_channel, stride=2)
        self.quantize_conv_b = Conv2d(embed_dim + channel, embed_dim, 1)
        self.quantize_b = Quantize(embed_dim, n_embed)
        self.upsample_t = nn.Sequential(ConvTranspose2d(embed_dim, embed_dim, 4, stride=2, padding=1))
        self.dec = Decoder(embed_dim + embed_dim, out_channel, channel, style_dim, n_res_block, n_res_channel, stride=6)
        self.embed_dim = 2 * embed_dim
        self._out_channel = out_channel

    def live_named_parameters(self):
        """Parameters that take part in training: everything but the dead `AdainResBlk.conv` stacks
        (vqvae_deep.py:120-125 builds them, :127-133 never calls them)."""
        dead = {id(p) for m in self.modules() if isinstance(m, AdainResBlk) for p in m.conv.parameters()}
        return [(k, p) for k, p in self.named_parameters() if id(p) not in dead]

    def live_parameters(self):
        return [p for _, p in self.live_named_parameters()]

    def encode(self, input):
        """-> (enc_b [B,C,H/8,W/8], enc_t [B,C,H/16,W/16])  (vqvae_deep.py:282-285)."""
        enc_b = self.enc_b(input)
        enc_t = self.enc_t(enc_b)
        return enc_b, enc_t

    def quantize(self, enc_b, enc_t):
        """-> (quant_t, quant_b, diff [1], id_t, id_b)  (vqvae_deep.py:287-301)."""
        quant_t, diff_t, id_t = self.quantize_t(self.quantize_conv_t.nhwc(ops.to_nhwc(enc_t)))
        dec_t = self.dec_t.nhwc(quant_t)
        cat = ops.CatFn.apply(dec_t, ops.to_nhwc(enc_b))
        quant_b, diff_b, id_b = self.quantize_b(self.quantize_conv_b.nhwc(cat))
        diff = ops.AddScalarsFn.apply(diff_t, diff_b)
        return quant_t.permute(0, 3, 1, 2), quant_b.permute(0, 3, 1, 2), diff, id_t, id_b

    def decode(self, quant, style):
        return self.dec(quant, style)

    def forward(self, input, style=None):
        if style is None:   # what the reference does at vqvae_deep.py:277: decode(quant) without its second argument
            raise TypeError("decode() missing 1 required positional argument: 'style'")
        enc_b, enc_t = self.encode(input)
        quant_t, quant_b, diff, _, _ = self.quantize(enc_b, enc_t)
        upsample_t = self.upsample_t(quant_t)
        quant = ops.CatFn.apply(ops.to_nhwc(upsample_t), ops.to_nhwc(quant_b)).permute(0, 3, 1, 2)
        dec = self.decode(quant, style)
        return dec, diff, quant

    def decode_code(self, code_t, code_b, style=None):
        """The fork passes (quant_t, quant_b) to decode(quant, style) (vqvae_deep.py:309-316), which cannot run; this
        is the intended composition: gather both codebooks, upsample the top level, concat, decode with `style`."""
        if style is None:
            raise TypeError("decode_code() needs the style tensor of the AdaIN decoder")
        quant_t = self.quantize_t.embed_code(code_t)
        quant_b = self.quantize_b.embed_code(code_b)
        up = self.upsample_t[0].nhwc(quant_t)
        quant = ops.CatFn.apply(up, quant_b)
        return ops.from_nhwc(self.dec.nhwc(quant, style), self._out_channel)
