"""Host-side mirror of the reference's vqvae.py module interface (same class names, constructor
signatures, forward contracts and state_dict keys/shapes) running on libvq2's HIP kernels.

  Quantize   <- /root/reference/vqvae.py:28-78
  ResBlock   <- vqvae.py:81-96
  Encoder    <- vqvae.py:99-127
  Decoder    <- vqvae.py:130-166
  VQVAE      <- vqvae.py:169-259

Module boundaries speak NCHW like the reference; inside, activations are NHWC and every
ReLU / bias / residual / concat is fused into a conv launch (see ops.py, csrc/vq2_conv.hip).
Results returned by a module are NCHW-*shaped* tensors (channels-last strides when C % 4 == 0,
so chaining modules never copies).
"""
import math

import torch
from torch import nn

from . import distributed as dist_fn
from . import ops
from .ops import ConvSpec


def _check_supported(spec):
    if spec.transposed:
        ok = spec.k == 4 and spec.stride == 2 and spec.pad == 1
    elif spec.stride == 1:
        ok = 1 <= spec.k <= 7 and 0 <= spec.pad < spec.k
    else:
        ok = spec.stride == 2 and spec.k == 4 and spec.pad == 1
    if not ok:
        raise NotImplementedError(
            f"vqvae2_amd implements the conv geometries of the VQ-VAE-2 path only "
            f"(stride-1 k<=7, k4/s2/p1, convT k4/s2/p1); got {spec}")


class _ConvBase(nn.Module):
    def reset_parameters(self):
        # same distribution as torch.nn.Conv2d / ConvTranspose2d default init
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = self.weight.shape[1] * self.weight.shape[2] * self.weight.shape[3]
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(self.bias, -bound, bound)

    def nhwc(self, x, relu_in=False, relu_out=False, residual=None, out=None, grad_stash=None, mask_input=False,
             premasked=False):
        return ops.conv_op(x, self.weight, self.bias, self.spec, relu_in, relu_out, residual, out, grad_stash,
                           mask_input, premasked)

    def forward(self, input):
        return ops.from_nhwc(self.nhwc(ops.to_nhwc(input)), self.spec.cout)


class Conv2d(_ConvBase):
    """nn.Conv2d(in, out, k, stride, padding) with the reference's parameter layout (OIHW)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.spec = ConvSpec(False, in_channels, out_channels, kernel_size, stride, padding)
        _check_supported(self.spec)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def extra_repr(self):
        s = self.spec
        return f"{s.cin}, {s.cout}, kernel_size={s.k}, stride={s.stride}, padding={s.pad}"


class ConvTranspose2d(_ConvBase):
    """nn.ConvTranspose2d(in, out, 4, stride=2, padding=1); weight layout (in, out, kh, kw)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.spec = ConvSpec(True, in_channels, out_channels, kernel_size, stride, padding)
        _check_supported(self.spec)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def extra_repr(self):
        s = self.spec
        return f"{s.cin}, {s.cout}, kernel_size={s.k}, stride={s.stride}, padding={s.pad}"


class ReLU(nn.Module):
    """Place-holder keeping nn.Sequential indices (and so state_dict keys) equal to the
    reference; inside Encoder/Decoder/ResBlock it is fused into the neighbouring conv."""

    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, input):
        x = ops.to_nhwc(input)
        return ops.from_nhwc(ops.ReluFn.apply(x), input.shape[1])


class Quantize(nn.Module):
    def __init__(self, dim, n_embed, decay=0.99, eps=1e-5):
        super().__init__()
        if dim not in (4, 8, 16, 32, 64, 128, 256) or n_embed % 4 != 0:
            raise NotImplementedError("vqvae2_amd.Quantize: dim must be a power of two in 4..256 and n_embed % 4 == 0")
        self.dim = dim
        self.n_embed = n_embed
        self.decay = decay
        self.eps = eps
        embed = torch.randn(dim, n_embed)
        self.register_buffer("embed", embed)
        self.register_buffer("cluster_size", torch.zeros(n_embed))
        self.register_buffer("embed_avg", embed.clone())
        # set by Stage1Trainer: EMA statistics go to a slice of the packed all-reduce buffer and the
        # update is applied after the collective (legal: the output uses the pre-update codebook)
        self.deferred_stats = None
        self.stats_stream = None     # with deferred_stats: side stream the statistics kernels may run on (see QuantizeFn)
        # prepared form of the codebook (embedT [K,D], ||e_k||^2 [K]) left by the EMA update's own launch; valid while
        # (storage, tensor version, number of raw-pointer updates) of `embed` are the ones it was made from
        self._prep = None
        self._prep_key = None
        self._raw_updates = 0

    def _embed_key(self):
        return (self.embed.data_ptr(), self.embed._version, self._raw_updates, self.embed.device)

    def _prepared(self):
        if self._prep is not None and self._prep_key == self._embed_key():
            return self._prep
        return None

    def invalidate_prepared(self):
        """Call after writing `embed` behind autograd's back (`embed.data.copy_()`, a raw-pointer collective, a foreign
        kernel): tensor._version does not move for those, so the cached (embedT, ||e||^2) would go stale silently."""
        self._prep = None
        self._prep_key = None
        self._raw_updates += 1

    def _ema_update(self, stats):
        # FRESH buffers every time: the previous pair may still be referenced by the autograd graph of the forward that
        # searched the pre-update codebook (the drop-in path updates inside forward, before backward reads embedT)
        self._prep = (torch.empty((self.n_embed, self.dim), device=self.embed.device, dtype=torch.float32),
                      torch.empty(self.n_embed, device=self.embed.device, dtype=torch.float32))
        fused = ops.vq_ema_update(self.embed, self.cluster_size, self.embed_avg, stats, self.decay, self.eps, self._prep)
        self._raw_updates += 1       # the kernels write through raw pointers: tensor._version does not move
        self._prep_key = self._embed_key() if fused else None

    def forward(self, input, _out=None):
        """input [B,H,W,dim] -> (quantize [B,H,W,dim], diff 0-dim, embed_ind [B,H,W] int64)."""
        if input.shape[-1] != self.dim:
            raise RuntimeError(f"Quantize: last dim {input.shape[-1]} != {self.dim}")
        x = input if input.dim() == 4 else input.reshape(-1, 1, 1, self.dim)
        want = self.training
        out, diff, ind, stats = ops.QuantizeFn.apply(x, self.embed, want, self.deferred_stats, _out, self._prepared(),
                                                      self.stats_stream if self.deferred_stats is not None else None)
        if want and self.deferred_stats is None:
            dist_fn.all_reduce(stats)  # counts and sums in ONE collective (vqvae.py:58-59 issues two)
            self._ema_update(stats)
        if input.dim() != 4:
            out = out.reshape(input.shape)
            ind = ind.reshape(input.shape[:-1])
        return out, diff, ind

    def apply_deferred_update(self):
        self._ema_update(self.deferred_stats)

    def embed_code(self, embed_id):
        return ops.vq_gather(embed_id, self.embed, self._prepared())


class ResBlock(nn.Module):
    def __init__(self, in_channel, channel):
        super().__init__()
        self.conv = nn.Sequential(
            ReLU(),
            Conv2d(in_channel, channel, 3, padding=1),
            ReLU(inplace=True),
            Conv2d(channel, in_channel, 1),
        )

    def nhwc(self, x, relu_in=False, relu_out=False, residual=None, out=None, premasked=False):
        assert not relu_in and residual is None
        c1, c2 = self.conv[1], self.conv[3]
        return ops.ResBlockFn.apply(x, c1.weight, c1.bias, c2.weight, c2.bias, c1.spec, c2.spec, relu_out, out,
                                    premasked)

    def forward(self, input):
        return ops.from_nhwc(self.nhwc(ops.to_nhwc(input)), input.shape[1])


def _run_blocks(blocks, x, out=None, grad_stash=None, mask_input=False, premasked=False):
    """Execute an nn.Sequential of {Conv2d, ConvTranspose2d, ReLU, ResBlock} with every ReLU fused
    into the conv that follows it (or, for a trailing ReLU, the op that precedes it)."""
    plan = []
    pending = False
    for m in blocks:
        if isinstance(m, ReLU):
            pending = True
        elif isinstance(m, ResBlock):
            if pending:
                plan.append(["relu", False, False])
            plan.append([m, False, False])
            pending = False
        elif isinstance(m, _ConvBase):
            plan.append([m, pending, False])
            pending = False
        else:
            raise NotImplementedError(f"unsupported block {type(m).__name__}")
    if pending:
        if plan and plan[-1][0] != "relu":
            plan[-1][2] = True
        else:
            plan.append(["relu", False, False])
    for i, (m, rin, rout) in enumerate(plan):
        last = i == len(plan) - 1
        if m == "relu":
            x = ops.ReluFn.apply(x)
            if last and out is not None:
                raise NotImplementedError("trailing stand-alone ReLU cannot target an output slice")
        else:
            kw = {}
            if i == 0 and isinstance(m, _ConvBase):
                kw.update(grad_stash=grad_stash, mask_input=mask_input)
            if last and rout and premasked:
                kw["premasked"] = True
            x = m.nhwc(x, relu_in=rin, relu_out=rout, out=out if last else None, **kw)
    first_is_conv = bool(plan) and isinstance(plan[0][0], _ConvBase)
    if (grad_stash is not None or mask_input) and not first_is_conv:
        raise NotImplementedError("grad_stash / mask_input need a conv as the first block")
    if premasked and not (plan and plan[-1][0] != "relu" and plan[-1][2]):
        raise NotImplementedError("premasked needs a trailing ReLU fused into the last block")
    return x


class Encoder(nn.Module):
    def __init__(self, in_channel, channel, n_res_block, n_res_channel, stride):
        super().__init__()
        if stride == 4:
            blocks = [
                Conv2d(in_channel, channel // 2, 4, stride=2, padding=1),
                ReLU(inplace=True),
                Conv2d(channel // 2, channel, 4, stride=2, padding=1),
                ReLU(inplace=True),
                Conv2d(channel, channel, 3, padding=1),
            ]
        elif stride == 2:
            blocks = [
                Conv2d(in_channel, channel // 2, 4, stride=2, padding=1),
                ReLU(inplace=True),
                Conv2d(channel // 2, channel, 3, padding=1),
            ]
        else:
            raise ValueError("stride must be 2 or 4")  # the reference leaves `blocks` undefined here
        for _ in range(n_res_block):
            blocks.append(ResBlock(channel, n_res_channel))
        blocks.append(ReLU(inplace=True))
        self.blocks = nn.Sequential(*blocks)
        self.out_channels = channel

    def nhwc(self, x, out=None, grad_stash=None, mask_input=False, premasked=False):
        """grad_stash / mask_input / premasked: see ops.GradStash and ops.ConvFn -- the trailing ReLU's backward
        mask and the fan-out gradient add are folded into the consumers' dgrad launches (training graph of VQVAE)."""
        return _run_blocks(self.blocks, x, out, grad_stash, mask_input, premasked)

    def forward(self, input):
        return ops.from_nhwc(self.nhwc(ops.to_nhwc(input)), self.out_channels)


class Decoder(nn.Module):
    def __init__(self, in_channel, out_channel, channel, n_res_block, n_res_channel, stride):
        super().__init__()
        blocks = [Conv2d(in_channel, channel, 3, padding=1)]
        for _ in range(n_res_block):
            blocks.append(ResBlock(channel, n_res_channel))
        blocks.append(ReLU(inplace=True))
        if stride == 4:
            blocks.extend([
                ConvTranspose2d(channel, channel // 2, 4, stride=2, padding=1),
                ReLU(inplace=True),
                ConvTranspose2d(channel // 2, out_channel, 4, stride=2, padding=1),
            ])
        elif stride == 2:
            blocks.append(ConvTranspose2d(channel, out_channel, 4, stride=2, padding=1))
        else:
            raise ValueError("stride must be 2 or 4")
        self.blocks = nn.Sequential(*blocks)
        self.out_channels = out_channel

    def nhwc(self, x, out=None):
        return _run_blocks(self.blocks, x, out)

    def forward(self, input):
        return ops.from_nhwc(self.nhwc(ops.to_nhwc(input)), self.out_channels)


class VQVAE(nn.Module):
    def __init__(self, in_channel=3, channel=128, n_res_block=2, n_res_channel=32, embed_dim=64, n_embed=512,
                 decay=0.99):
        super().__init__()
        self.enc_b = Encoder(in_channel, channel, n_res_block, n_res_channel, stride=4)
        self.enc_t = Encoder(channel, channel, n_res_block, n_res_channel, stride=2)
        self.quantize_conv_t = Conv2d(channel, embed_dim, 1)
        self.quantize_t = Quantize(embed_dim, n_embed)  # `decay` is not forwarded (vqvae.py:185)
        self.dec_t = Decoder(embed_dim, embed_dim, channel, n_res_block, n_res_channel, stride=2)
        self.quantize_conv_b = Conv2d(embed_dim + channel, embed_dim, 1)
        self.quantize_b = Quantize(embed_dim, n_embed)
        self.upsample_t = ConvTranspose2d(embed_dim, embed_dim, 4, stride=2, padding=1)
        self.dec = Decoder(embed_dim + embed_dim, in_channel, channel, n_res_block, n_res_channel, stride=4)
        # never executed by the reference either (vqvae.py:203-210); kept for state_dict parity
        self.dec_ir = Decoder(embed_dim + embed_dim, 1, channel, n_res_block + 2, n_res_channel, stride=4)
        self.embed_dim = 2 * embed_dim
        self._in_channel, self._channel, self._e = in_channel, channel, embed_dim

    # ---- parameters that take part in training (dec_ir never receives a gradient)
    def live_named_parameters(self):
        return [(k, p) for k, p in self.named_parameters() if not k.startswith("dec_ir.")]

    def live_parameters(self):
        return [p for _, p in self.live_named_parameters()]

    # ---- internal NHWC graph; torch.cat of vqvae.py:233 / :218 becomes channel-slice outputs
    def _encode_nhwc(self, x, quant_b_out=None):
        e, c = self._e, self._channel
        n, h, w, _ = x.shape
        if h % 8 or w % 8:
            raise RuntimeError("VQVAE: input height/width must be multiples of 8")
        cat = torch.empty((n, h // 4, w // 4, e + c), device=x.device, dtype=torch.float32)
        # enc_b feeds enc_t AND the concat (vqvae.py:225,233): the concat's gradient slice is added inside
        # the dgrad launch of enc_t's first conv (ops.GradStash) instead of by an add kernel, and that launch
        # also applies the backward mask of enc_b's trailing ReLU (vqvae.py:122) to the sum; likewise
        # quantize_conv_t's dgrad masks for enc_t's trailing ReLU -- no stand-alone ReLU-backward passes
        fuse = torch.is_grad_enabled()
        skip = ops.GradStash(strict=True) if fuse else None
        enc_b = self.enc_b.nhwc(x, out=cat[..., e:], premasked=fuse)
        enc_t = self.enc_t.nhwc(enc_b, grad_stash=skip, mask_input=fuse, premasked=fuse)
        quant_t, diff_t, id_t = self.quantize_t(self.quantize_conv_t.nhwc(enc_t, mask_input=fuse))
        quant_t_for_dec, quant_t_ret = ops.FanOutFn.apply(quant_t)
        dec_t = self.dec_t.nhwc(quant_t_for_dec, out=cat[..., :e])
        enc_cat = ops.CatViewFn.apply(dec_t, enc_b, cat, skip)
        quant_b, diff_b, id_b = self.quantize_b(self.quantize_conv_b.nhwc(enc_cat), _out=quant_b_out)
        diff = ops.AddScalarsFn.apply(diff_t, diff_b)
        return quant_t_ret, quant_b, diff, id_t, id_b

    def encode(self, input):
        quant_t, quant_b, diff, id_t, id_b = self._encode_nhwc(ops.to_nhwc(input))
        return quant_t.permute(0, 3, 1, 2), quant_b.permute(0, 3, 1, 2), diff, id_t, id_b

    def _decode_from(self, quant_t, cat, quant_b):
        e = self._e
        up = self.upsample_t.nhwc(quant_t, out=cat[..., :e])
        quant = ops.CatViewFn.apply(up, quant_b, cat)
        return self.dec.nhwc(quant)

    def forward_nhwc(self, x):
        """NHWC in ([N,H,W,ceil4(in_channel)], zero padded) -> (NHWC reconstruction, diff [1])."""
        n, h, w, _ = x.shape
        e = self._e
        cat = torch.empty((n, h // 4, w // 4, 2 * e), device=x.device, dtype=torch.float32)
        quant_t, quant_b, diff, _, _ = self._encode_nhwc(x, quant_b_out=cat[..., e:])
        return self._decode_from(quant_t, cat, quant_b), diff

    def forward(self, input):
        dec, diff = self.forward_nhwc(ops.to_nhwc(input))
        return ops.from_nhwc(dec, self._in_channel), diff

    def decode(self, quant):
        return self.dec(quant)

    def decode_code(self, code_t, code_b):
        """Upstream semantics (the fork's version calls decode() with two arguments and cannot
        run, vqvae.py:257): gather both codebooks, upsample the top level, concat, decode."""
        e = self._e
        quant_t = self.quantize_t.embed_code(code_t)
        n, hb, wb = code_b.shape
        cat = torch.empty((n, hb, wb, 2 * e), device=quant_t.device, dtype=torch.float32)
        qb = self.quantize_b.embed_code(code_b)
        ops.check(ops.lib.vq2_slice_copy(ops._p(qb), e, ops._p(cat[..., e:]), 2 * e, n * hb * wb, e, 0, ops._stream()),
                  "slice_copy")
        up = self.upsample_t.nhwc(quant_t, out=cat[..., :e])
        del up
        dec = self.dec.nhwc(cat)
        return ops.from_nhwc(dec, self._in_channel)
