"""GPU edge cases of the C-ABI path: ragged sizes, channel slices (pixel stride > C), fused flags,
layout conversion, stand-alone modules, optimizer and checkpoint round trip -- each against the
CPU oracle ops (plain torch fp32 on the same seeded inputs)."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rng
from oracle import vqvae_oracle as O

pytestmark = pytest.mark.gpu
RT, AT = 2e-4, 2e-5


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=RT, atol=AT, what=""):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol, err_msg=what)


@pytest.fixture(scope="module")
def amd():
    import vqvae2_amd
    return vqvae2_amd


RAGGED = [
    # kind, cin, cout, k, stride, pad, N, H, W
    ("conv", 8, 12, 3, 1, 1, 1, 5, 7), ("conv", 20, 36, 3, 1, 1, 3, 9, 4), ("conv", 4, 8, 1, 1, 0, 2, 1, 1),
    ("conv", 12, 8, 5, 1, 2, 1, 6, 6), ("conv", 16, 40, 7, 1, 3, 1, 8, 5), ("conv", 3, 8, 4, 2, 1, 2, 10, 6),
    ("conv", 36, 132, 4, 2, 1, 1, 12, 8), ("conv", 5, 7, 3, 1, 1, 2, 4, 4), ("conv", 160, 136, 3, 1, 1, 1, 13, 11),
    ("convT", 8, 12, 4, 2, 1, 1, 3, 5), ("convT", 20, 3, 4, 2, 1, 2, 5, 3), ("convT", 16, 2, 4, 2, 1, 1, 4, 70),
    ("convT", 132, 68, 4, 2, 1, 1, 6, 7), ("convT", 5, 6, 4, 2, 1, 1, 2, 2),
    # shapes that take the LDS-patch sub-pixel kernel (output channels % 64 == 0, reduction channels % 16 == 0):
    # ragged tiles, two 64-channel groups, and the data gradient of a stride-2 conv
    ("convT", 32, 64, 4, 2, 1, 2, 5, 19), ("convT", 16, 128, 4, 2, 1, 1, 9, 33), ("conv", 64, 48, 4, 2, 1, 2, 10, 14),
    ("conv", 128, 32, 4, 2, 1, 1, 18, 34),
    # the <= 3-channel reconstruction kernel (32-channel slices): ragged 8x32 tiles
    ("convT", 32, 3, 4, 2, 1, 2, 9, 35), ("convT", 64, 2, 4, 2, 1, 1, 17, 5),
]


def test_ragged_conv_shapes_fwd_and_grads(amd):
    dev = torch.device("cuda:0")
    for idx, (kind, cin, cout, k, s, p, n, h, w) in enumerate(RAGGED):
        tag = f"rag{idx}"
        x = t(rng.normal(3, tag + ".x", (n, cin, h, w)))
        if kind == "conv":
            m = amd.Conv2d(cin, cout, k, stride=s, padding=p)
            wshape = (cout, cin, k, k)
        else:
            m = amd.ConvTranspose2d(cin, cout, k, stride=s, padding=p)
            wshape = (cin, cout, k, k)
        wt = t(rng.uniform(3, tag + ".w", wshape, -0.3, 0.3))
        b = t(rng.uniform(3, tag + ".b", (cout,), -1, 1))
        m.load_state_dict({"weight": wt, "bias": b})
        m.to(dev)
        xr = x.clone().requires_grad_(True)
        wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        fn = F.conv2d if kind == "conv" else F.conv_transpose2d
        yr = fn(xr, wr, br, stride=s, padding=p)
        gy = t(rng.normal(3, tag + ".gy", tuple(yr.shape)))
        yr.backward(gy)
        xg = x.to(dev).requires_grad_(True)
        y = m(xg)
        assert tuple(y.shape) == tuple(yr.shape), tag
        y.backward(gy.to(dev))
        close(y, yr, what=tag + ".y")
        close(xg.grad, xr.grad, what=tag + ".gx")
        close(m.weight.grad, wr.grad, rtol=5e-4, atol=1e-4, what=tag + ".gw")
        close(m.bias.grad, br.grad, rtol=5e-4, atol=1e-4, what=tag + ".gb")


def test_channel_slices_and_fused_flags(amd):
    """conv reading a channel slice, writing into a channel slice of a wider buffer, with fused
    ReLU-in / residual / ReLU-out (what torch.cat + ResBlock become), against unfused torch ops."""
    from vqvae2_amd import ops
    dev = torch.device("cuda:0")
    n, h, w, ci, co = 2, 6, 5, 12, 8
    wide_in = t(rng.normal(5, "sl.in", (n, h, w, 20))).to(dev)
    x = wide_in[..., 4:16]                                   # pixel stride 20, 12 channels
    wide_out = torch.full((n, h, w, 24), 7.0, device=dev)
    res = t(rng.normal(5, "sl.res", (n, h, w, co))).to(dev)
    wt = t(rng.uniform(5, "sl.w", (co, ci, 3, 3), -0.3, 0.3)).to(dev)
    b = t(rng.uniform(5, "sl.b", (co,), -1, 1)).to(dev)
    spec = ops.ConvSpec(False, ci, co, 3, 1, 1)
    y = ops.conv_forward(spec, x, wt, b, ops.VQ2_RELU_IN | ops.VQ2_RELU_OUT, residual=res, out=wide_out[..., 8:16])
    ref = F.relu(F.conv2d(F.relu(x.permute(0, 3, 1, 2).cpu()), wt.cpu(), b.cpu(), padding=1)
                 + res.permute(0, 3, 1, 2).cpu()).permute(0, 2, 3, 1)
    close(y, ref)
    assert float(wide_out[..., :8].min()) == 7.0 and float(wide_out[..., 16:].max()) == 7.0   # neighbours untouched
    # data gradient with fused ReLU mask + skip gradient, written into a slice
    dy = t(rng.normal(5, "sl.dy", (n, h, w, co))).to(dev)
    dx_wide = torch.zeros((n, h, w, 16), device=dev)
    dx = ops.conv_dgrad(spec, (n, h, w, ci), dy, wt, mask=x, residual=wide_in[..., 0:12], out=dx_wide[..., 4:16])
    xr = x.permute(0, 3, 1, 2).cpu().clone().requires_grad_(True)
    F.conv2d(F.relu(xr), wt.cpu(), None, padding=1).backward(dy.permute(0, 3, 1, 2).cpu())
    close(dx, xr.grad.permute(0, 2, 3, 1) + wide_in[..., 0:12].cpu())


def _launched(amd, fn):
    """Run fn with every instrumented launch bracketed; returns (result, kernel labels seen)."""
    import ctypes
    lib = amd._lib.lib
    lib.vq2_prof_enable(1)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        lib.vq2_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.vq2_prof_report(buf, len(buf))
    return out, [ln.split()[0] for ln in buf.value.decode().splitlines()]


@pytest.mark.parametrize("shape", [(2, 6, 64, 64, 128), (1, 4, 128, 72, 256), (3, 2, 64, 128, 128), (2, 6, 128, 128, 256),
                                   (2, 8, 32, 64, 128), (1, 4, 32, 128, 64), (2, 4, 64, 64, 64), (33, 32, 32, 64, 128), (2, 6, 64, 32, 128),
                                   (3, 8, 32, 128, 128)])
def test_winograd_rows_conv_forward_and_data_gradient(amd, shape):
    """3x3 stride-1 convolutions with >= 32 input channels, whole 64-channel output tiles and rows of whole 64-pixel segments
    (or 32-pixel ones, H % 4 == 0: the 32x32 level; the last case is a launch large enough for 128-wide tiles there) run as
    F(2,3) Winograd along the rows (csrc/vq2_wino.hip): forward with ReLU-in / bias / residual / ReLU-out
    through channel slices, data gradient with the ReLU mask and the skip gradient, against fp64 torch on the CPU.  The
    image border (zero padding on all four sides) and the seams between 64-pixel segments are part of every case."""
    from vqvae2_amd import ops
    dev = torch.device("cuda:0")
    n, h, w, ci, co = shape
    tag = "wino%dx%dx%d" % (h, w, ci)
    wide_in = t(rng.normal(9, tag + ".in", (n, h, w, ci + 8))).to(dev)
    x = wide_in[..., 4:4 + ci]                                # pixel stride ci + 8
    wide_out = torch.full((n, h, w, co + 8), 7.0, device=dev)
    res = t(rng.normal(9, tag + ".res", (n, h, w, co))).to(dev)
    wt = t(rng.uniform(9, tag + ".w", (co, ci, 3, 3), -0.1, 0.1)).to(dev)
    b = t(rng.uniform(9, tag + ".b", (co,), -1, 1)).to(dev)
    spec = ops.ConvSpec(False, ci, co, 3, 1, 1)
    y, seen = _launched(amd, lambda: ops.conv_forward(spec, x, wt, b, ops.VQ2_RELU_IN | ops.VQ2_RELU_OUT, residual=res,
                                                      out=wide_out[..., 4:4 + co]))
    assert any(k.startswith("conv_wino3") for k in seen), seen
    x64 = x.permute(0, 3, 1, 2).cpu().double()
    ref = F.relu(F.conv2d(F.relu(x64), wt.cpu().double(), b.cpu().double(), padding=1)
                 + res.permute(0, 3, 1, 2).cpu().double()).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    close(y.double(), ref, rtol=0, atol=5e-6 * scale, what=tag + ".y")       # fp32 rounding of a depth-9*ci sum
    assert float(wide_out[..., :4].min()) == 7.0 and float(wide_out[..., 4 + co:].max()) == 7.0
    y2, seen = _launched(amd, lambda: ops.conv_forward(spec, x, wt, None, 0))   # no bias, no ReLU, dense output
    assert any(k.startswith("conv_wino3") for k in seen), seen
    ref2 = F.conv2d(x64, wt.cpu().double(), None, padding=1).permute(0, 2, 3, 1)
    close(y2.double(), ref2, rtol=0, atol=5e-6 * float(ref2.abs().max()), what=tag + ".y2")
    # data gradient (the same kernel on the flipped panel) needs whole 64-channel tiles of the INPUT channel count
    if ci % 64 == 0:
        dy = t(rng.normal(9, tag + ".dy", (n, h, w, co))).to(dev)
        dx_wide = torch.zeros((n, h, w, ci + 8), device=dev)
        dx, seen = _launched(amd, lambda: ops.conv_dgrad(spec, (n, h, w, ci), dy, wt, mask=x, residual=wide_in[..., 0:ci],
                                                          out=dx_wide[..., 4:4 + ci]))
        assert any(k.startswith("conv_wino3") for k in seen), seen
        xr = x64.clone().requires_grad_(True)
        F.conv2d(F.relu(xr), wt.cpu().double(), None, padding=1).backward(dy.permute(0, 3, 1, 2).cpu().double())
        refg = xr.grad.permute(0, 2, 3, 1) + wide_in[..., 0:ci].cpu().double()
        close(dx.double(), refg, rtol=0, atol=5e-6 * float(refg.abs().max()), what=tag + ".dx")
        # weight and bias gradient: the same transform on the reduction side (wgrad_fast_kernel<..., WINO>, whole 128-channel
        # tiles on both sides, rows of whole 64-pixel segments), with and without the fused ReLU on x, through channel slices
        if co % 128 == 0 and ci % 128 == 0 and (w % 64 == 0 or (w == 32 and h % 2 == 0)):
            wr = wt.cpu().double().clone().requires_grad_(True)
            br = b.cpu().double().clone().requires_grad_(True)
            for relu_in in (True, False):
                (dw, db), seen = _launched(amd, lambda: ops.conv_wgrad(spec, x, dy, relu_in, wt, b))
                assert any("wino" in k and k.startswith("wgrad") for k in seen), seen
                wr.grad = br.grad = None
                F.conv2d(F.relu(x64) if relu_in else x64, wr, br, padding=1).backward(dy.permute(0, 3, 1, 2).cpu().double())
                close(dw.double(), wr.grad, rtol=0, atol=1e-5 * float(wr.grad.abs().max()), what=tag + ".dw%d" % relu_in)
                close(db.double(), br.grad, rtol=0, atol=1e-5 * float(br.grad.abs().max()), what=tag + ".db%d" % relu_in)


@pytest.mark.parametrize("shape", [(32, 32, 128, 64, 256), (1, 4, 256, 40, 256), (2, 16, 64, 64, 128), (2, 8, 128, 64, 64),
                                   (13, 64, 64, 32, 512), (3, 12, 128, 32, 128)])
def test_winograd_by_parity_k4s2_conv_forward_and_convT_data_gradient(amd, shape):
    """4x4 stride-2 convolutions with whole 64-channel output tiles and output rows of whole 64-pixel segments (or 32-pixel
    ones with Ho % 4 == 0 and whole 128-channel tiles; the first and the last case are large enough for 128-wide tiles) run
    as two F(2,2) filters by column parity (csrc/vq2_wino.hip): the forward with ReLU-in / bias / ReLU-out through channel
    slices, and the data gradient of ConvTranspose2d(k4,s2,p1) (the same operation on dy) with its ReLU mask, vs fp64."""
    from vqvae2_amd import ops
    dev = torch.device("cuda:0")
    n, h, w, ci, co = shape                                   # input h x w, output h/2 x w/2
    tag = "k4s2_%dx%dx%d" % (h, w, ci)
    wide_in = t(rng.normal(11, tag + ".in", (n, h, w, ci + 4))).to(dev)
    x = wide_in[..., 4:4 + ci]
    wide_out = torch.full((n, h // 2, w // 2, co + 8), 7.0, device=dev)
    wt = t(rng.uniform(11, tag + ".w", (co, ci, 4, 4), -0.1, 0.1)).to(dev)
    b = t(rng.uniform(11, tag + ".b", (co,), -1, 1)).to(dev)
    spec = ops.ConvSpec(False, ci, co, 4, 2, 1)
    y, seen = _launched(amd, lambda: ops.conv_forward(spec, x, wt, b, ops.VQ2_RELU_IN | ops.VQ2_RELU_OUT,
                                                      out=wide_out[..., 4:4 + co]))
    assert any(k.startswith("conv_wino_k4s2") for k in seen), seen
    x64 = x.permute(0, 3, 1, 2).cpu().double()
    ref = F.relu(F.conv2d(F.relu(x64), wt.cpu().double(), b.cpu().double(), stride=2, padding=1)).permute(0, 2, 3, 1)
    close(y.double(), ref, rtol=0, atol=5e-6 * float(ref.abs().max()), what=tag + ".y")
    assert float(wide_out[..., :4].min()) == 7.0 and float(wide_out[..., 4 + co:].max()) == 7.0
    # ConvTranspose2d(co -> ci) backward: dx[n, h/2, w/2, co] = strided conv of dy[n, h, w, ci] with the same taps
    tspec = ops.ConvSpec(True, co, ci, 4, 2, 1)
    wtt = t(rng.uniform(11, tag + ".wt", (co, ci, 4, 4), -0.1, 0.1)).to(dev)     # IOHW of the transposed conv
    xin = t(rng.normal(11, tag + ".xin", (n, h // 2, w // 2, co))).to(dev)
    dy = t(rng.normal(11, tag + ".dy", (n, h, w, ci))).to(dev)
    dx, seen = _launched(amd, lambda: ops.conv_dgrad(tspec, (n, h // 2, w // 2, co), dy, wtt, mask=xin))
    assert any(k.startswith("conv_wino_k4s2") for k in seen), seen
    xr = xin.permute(0, 3, 1, 2).cpu().double().clone().requires_grad_(True)
    F.conv_transpose2d(F.relu(xr), wtt.cpu().double(), None, stride=2, padding=1).backward(dy.permute(0, 3, 1, 2).cpu().double())
    refg = xr.grad.permute(0, 2, 3, 1)
    close(dx.double(), refg, rtol=0, atol=5e-6 * float(refg.abs().max()), what=tag + ".dx")
    # weight / bias gradients of both layer kinds (64 gathered channels, whole 128-channel tiles): F(2,2) over column pairs
    if ci == 64 and co % 128 == 0 and ((w // 2) % 64 == 0 or (w // 2 == 32 and (h // 2) % 2 == 0)):
        gy = t(rng.normal(11, tag + ".gy", (n, h // 2, w // 2, co))).to(dev)
        wr = wt.cpu().double().clone().requires_grad_(True)
        br = b.cpu().double().clone().requires_grad_(True)
        for relu_in in (True, False):
            (dw, db), seen = _launched(amd, lambda: ops.conv_wgrad(spec, x, gy, relu_in, wt, b))
            assert any(k.startswith("wgrad") and "wino4" in k for k in seen), seen
            wr.grad = br.grad = None
            F.conv2d(F.relu(x64) if relu_in else x64, wr, br, stride=2, padding=1).backward(gy.permute(0, 3, 1, 2).cpu().double())
            close(dw.double(), wr.grad, rtol=0, atol=1e-5 * float(wr.grad.abs().max()), what=tag + ".dw%d" % relu_in)
            close(db.double(), br.grad, rtol=0, atol=1e-5 * float(br.grad.abs().max()), what=tag + ".db%d" % relu_in)
        bt = t(rng.uniform(11, tag + ".bt", (ci,), -1, 1)).to(dev)
        wtr = wtt.cpu().double().clone().requires_grad_(True)
        btr = bt.cpu().double().clone().requires_grad_(True)
        xin64 = xin.permute(0, 3, 1, 2).cpu().double()
        for relu_in in (True, False):
            (dw, db), seen = _launched(amd, lambda: ops.conv_wgrad(tspec, xin, dy, relu_in, wtt, bt))
            assert any(k.startswith("wgrad") and "wino4" in k for k in seen), seen
            wtr.grad = btr.grad = None
            F.conv_transpose2d(F.relu(xin64) if relu_in else xin64, wtr, btr, stride=2,
                               padding=1).backward(dy.permute(0, 3, 1, 2).cpu().double())
            close(dw.double(), wtr.grad, rtol=0, atol=1e-5 * float(wtr.grad.abs().max()), what=tag + ".dwt%d" % relu_in)
            close(db.double(), btr.grad, rtol=0, atol=1e-5 * float(btr.grad.abs().max()), what=tag + ".dbt%d" % relu_in)


@pytest.mark.parametrize("shape", [(2, 6, 64, 128), (1, 5, 128, 128), (3, 3, 64, 256), (2, 6, 32, 128), (3, 4, 32, 256)])
def test_winograd_exchanged_roles_weight_gradient_of_the_resblock_3x3(amd, shape):
    """The 3x3 conv into 32 channels (ResBlock, vqvae.py:87) has its weight gradient computed with the roles of x and dy
    exchanged; on rows of whole 64-pixel segments that sum runs as F(2,3) over column pairs (wgrad_fast_kernel<..., 3>).
    dw and db (taken from the gathered dy operand) with and without the fused ReLU on x, through channel slices, vs fp64."""
    from vqvae2_amd import ops
    dev = torch.device("cuda:0")
    n, h, w, ci = shape
    co = 32
    tag = "swwino%dx%dx%d" % (h, w, ci)
    wide_in = t(rng.normal(13, tag + ".in", (n, h, w, ci + 8))).to(dev)
    x = wide_in[..., 4:4 + ci]
    wide_dy = t(rng.normal(13, tag + ".dy", (n, h, w, co + 4))).to(dev)
    dy = wide_dy[..., 4:4 + co]
    wt = t(rng.uniform(13, tag + ".w", (co, ci, 3, 3), -0.1, 0.1)).to(dev)
    b = t(rng.uniform(13, tag + ".b", (co,), -1, 1)).to(dev)
    spec = ops.ConvSpec(False, ci, co, 3, 1, 1)
    x64 = x.permute(0, 3, 1, 2).cpu().double()
    wr = wt.cpu().double().clone().requires_grad_(True)
    br = b.cpu().double().clone().requires_grad_(True)
    for relu_in in (True, False):
        (dw, db), seen = _launched(amd, lambda: ops.conv_wgrad(spec, x, dy, relu_in, wt, b))
        assert any(k.startswith("wgrad") and "swwino" in k for k in seen), seen
        wr.grad = br.grad = None
        F.conv2d(F.relu(x64) if relu_in else x64, wr, br, padding=1).backward(dy.permute(0, 3, 1, 2).cpu().double())
        close(dw.double(), wr.grad, rtol=0, atol=1e-5 * float(wr.grad.abs().max()), what=tag + ".dw%d" % relu_in)
        close(db.double(), br.grad, rtol=0, atol=1e-5 * float(br.grad.abs().max()), what=tag + ".db%d" % relu_in)


@pytest.mark.parametrize("shape", [(2, 8, 64, 128, 64), (1, 4, 128, 40, 128), (3, 12, 64, 32, 64), (2, 8, 32, 128, 64),
                                   (3, 16, 32, 64, 128)])
def test_winograd_subpixel_conv_transpose_forward_and_k4s2_data_gradient(amd, shape):
    """ConvTranspose2d(k4,s2,p1) with >= 32 input channels, output channels in whole 64-tiles and input rows of whole
    64-pixel segments: every output phase is a 2-tap filter along the row and runs as F(2,2) (wino_subpixel_kernel).
    Forward with ReLU-in / bias / ReLU-out through channel slices, and the data gradient of the 4x4 stride-2 conv (the same
    operation on dy) with its ReLU mask, against fp64."""
    from vqvae2_amd import ops
    dev = torch.device("cuda:0")
    n, h, w, ci, co = shape                                   # input h x w, output 2h x 2w
    tag = "spw_%dx%dx%d" % (h, w, ci)
    wide_in = t(rng.normal(17, tag + ".in", (n, h, w, ci + 4))).to(dev)
    x = wide_in[..., 4:4 + ci]
    wide_out = torch.full((n, 2 * h, 2 * w, co + 8), 7.0, device=dev)
    wt = t(rng.uniform(17, tag + ".w", (ci, co, 4, 4), -0.1, 0.1)).to(dev)      # IOHW
    b = t(rng.uniform(17, tag + ".b", (co,), -1, 1)).to(dev)
    tspec = ops.ConvSpec(True, ci, co, 4, 2, 1)
    y, seen = _launched(amd, lambda: ops.conv_forward(tspec, x, wt, b, ops.VQ2_RELU_IN | ops.VQ2_RELU_OUT,
                                                      out=wide_out[..., 4:4 + co]))
    assert any(k.startswith("conv_wino_subpixel") for k in seen), seen
    x64 = x.permute(0, 3, 1, 2).cpu().double()
    ref = F.relu(F.conv_transpose2d(F.relu(x64), wt.cpu().double(), b.cpu().double(), stride=2, padding=1)).permute(0, 2, 3, 1)
    close(y.double(), ref, rtol=0, atol=5e-6 * float(ref.abs().max()), what=tag + ".y")
    assert float(wide_out[..., :4].min()) == 7.0 and float(wide_out[..., 4 + co:].max()) == 7.0
    # data gradient of Conv2d(co -> ci, k4 s2 p1) whose input is the 2h x 2w tensor: dx = conv_transpose(dy)
    spec = ops.ConvSpec(False, co, ci, 4, 2, 1)
    wc = t(rng.uniform(17, tag + ".wc", (ci, co, 4, 4), -0.1, 0.1)).to(dev)      # OIHW of that conv: [ci][co]
    xin = t(rng.normal(17, tag + ".xin", (n, 2 * h, 2 * w, co))).to(dev)
    dy = t(rng.normal(17, tag + ".dy", (n, h, w, ci))).to(dev)
    dx, seen = _launched(amd, lambda: ops.conv_dgrad(spec, (n, 2 * h, 2 * w, co), dy, wc, mask=xin))
    assert any(k.startswith("conv_wino_subpixel") for k in seen), seen
    xr = xin.permute(0, 3, 1, 2).cpu().double().clone().requires_grad_(True)
    F.conv2d(F.relu(xr), wc.cpu().double(), None, stride=2, padding=1).backward(dy.permute(0, 3, 1, 2).cpu().double())
    refg = xr.grad.permute(0, 2, 3, 1)
    close(dx.double(), refg, rtol=0, atol=5e-6 * float(refg.abs().max()), what=tag + ".dx")


@pytest.mark.parametrize("shape", [(4, 64, 64, 192), (5, 60, 61, 128), (3, 80, 70, 64)])
def test_row_streaming_1x1_conv_out_of_64_channels(amd, shape):
    """1x1 convolutions with exactly 64 input channels and 64..192 output channels on >= 16,384 pixels (the data gradient of
    quantize_conv_b, vqvae.py:189, and the other layers fed by embed_dim) run on the row-streaming kernel
    (conv1x1_k64_kernel): forward with ReLU-in / bias / residual / ReLU-out through channel slices, and as the data gradient
    of a 1x1 conv into 64 channels with its ReLU mask; pixel counts that are not multiples of 32 included."""
    from vqvae2_amd import ops
    dev = torch.device("cuda:0")
    n, h, w, co = shape
    ci = 64
    tag = "k64_%dx%dx%d" % (h, w, co)
    wide_in = t(rng.normal(19, tag + ".in", (n, h, w, ci + 8))).to(dev)
    x = wide_in[..., 4:4 + ci]
    wide_out = torch.full((n, h, w, co + 8), 7.0, device=dev)
    res = t(rng.normal(19, tag + ".res", (n, h, w, co))).to(dev)
    wt = t(rng.uniform(19, tag + ".w", (co, ci, 1, 1), -0.2, 0.2)).to(dev)
    b = t(rng.uniform(19, tag + ".b", (co,), -1, 1)).to(dev)
    spec = ops.ConvSpec(False, ci, co, 1, 1, 0)
    y, seen = _launched(amd, lambda: ops.conv_forward(spec, x, wt, b, ops.VQ2_RELU_IN | ops.VQ2_RELU_OUT, residual=res,
                                                      out=wide_out[..., 4:4 + co]))
    assert any(k.startswith("conv1x1_k64") for k in seen), seen
    x64 = x.permute(0, 3, 1, 2).cpu().double()
    ref = F.relu(F.conv2d(F.relu(x64), wt.cpu().double(), b.cpu().double()) + res.permute(0, 3, 1, 2).cpu().double()).permute(0, 2, 3, 1)
    close(y.double(), ref, rtol=0, atol=2e-6 * float(ref.abs().max()), what=tag + ".y")
    assert float(wide_out[..., :4].min()) == 7.0 and float(wide_out[..., 4 + co:].max()) == 7.0
    # data gradient of Conv2d(co -> 64, 1x1): dx[.., co] = dy[.., 64] W, masked by the layer's pre-ReLU input
    dspec = ops.ConvSpec(False, co, ci, 1, 1, 0)
    wd = t(rng.uniform(19, tag + ".wd", (ci, co, 1, 1), -0.2, 0.2)).to(dev)
    xin = t(rng.normal(19, tag + ".xin", (n, h, w, co))).to(dev)
    dy = t(rng.normal(19, tag + ".dy", (n, h, w, ci))).to(dev)
    dx, seen = _launched(amd, lambda: ops.conv_dgrad(dspec, (n, h, w, co), dy, wd, mask=xin))
    assert any(k.startswith("conv1x1_k64") for k in seen), seen
    xr = xin.permute(0, 3, 1, 2).cpu().double().clone().requires_grad_(True)
    F.conv2d(F.relu(xr), wd.cpu().double(), None).backward(dy.permute(0, 3, 1, 2).cpu().double())
    refg = xr.grad.permute(0, 2, 3, 1)
    close(dx.double(), refg, rtol=0, atol=2e-6 * float(refg.abs().max()), what=tag + ".dx")


def test_layout_conversion_generic_channels(amd):
    from vqvae2_amd import ops
    dev = torch.device("cuda:0")
    for c in (1, 3, 4, 6, 33, 130):
        x = t(rng.normal(9, f"lay{c}", (2, c, 5, 37))).to(dev).requires_grad_(True)
        y = ops.NchwToNhwc.apply(x)
        assert y.shape[-1] == ops.ceil4(c)
        close(y[..., :c], x.permute(0, 2, 3, 1), rtol=0, atol=0)
        assert float(y.detach()[..., c:].abs().sum()) == 0.0
        back = ops.NhwcToNchw.apply(y, c)
        close(back, x, rtol=0, atol=0)
        back.sum().backward()
        close(x.grad, torch.ones_like(x), rtol=0, atol=0)


def test_standalone_modules_and_channels_last_chaining(amd):
    dev = torch.device("cuda:0")
    x = t(rng.normal(4, "sm.x", (2, 16, 6, 6))).to(dev)
    r = amd.ReLU()(x)
    close(r, F.relu(x), rtol=0, atol=0)
    c1 = amd.Conv2d(16, 32, 3, padding=1).to(dev)
    c2 = amd.Conv2d(32, 8, 1).to(dev)
    y1 = c1(x)                                               # NCHW-shaped, channels-last strides
    assert tuple(y1.shape) == (2, 32, 6, 6) and y1.stride(1) == 1
    y2 = c2(y1)                                              # consumed zero-copy
    ref = F.conv2d(F.conv2d(x.cpu(), c1.weight.detach().cpu(), c1.bias.detach().cpu(), padding=1),
                   c2.weight.detach().cpu(), c2.bias.detach().cpu())
    close(y2, ref)
    close(y2.contiguous(), ref)
    q = amd.Quantize(16, 64).to(dev).eval()
    flat = t(rng.normal(4, "sm.q", (10, 16))).to(dev)       # non-4D input, like any [..., D] tensor
    out, diff, idx = q(flat)
    assert tuple(out.shape) == (10, 16) and tuple(idx.shape) == (10,)
    ro, rd, ri = O.quantize_forward(flat.cpu(), q.embed.cpu().clone(), torch.zeros(64), q.embed.cpu().clone(), False)
    assert torch.equal(idx.cpu(), ri)
    close(out, ro, rtol=1e-6, atol=1e-6)


def test_fused_adam_matches_torch_adam(amd):
    dev = torch.device("cuda:0")
    ps = [torch.nn.Parameter(t(rng.normal(6, f"ad{i}", s)).to(dev)) for i, s in enumerate([(7,), (3, 5), (2, 3, 4, 4)])]
    qs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    a = amd.FusedAdam(ps, lr=1e-2)
    b = torch.optim.Adam(qs, lr=1e-2)
    for step in range(4):
        for i, (p, q) in enumerate(zip(ps, qs)):
            g = t(rng.normal(6, f"adg{step}.{i}", tuple(p.shape)))
            p.grad = g.to(dev)
            q.grad = g.clone()
        a.step()
        b.step()
    for p, q in zip(ps, qs):
        close(p, q, rtol=1e-5, atol=1e-6)


def test_checkpoint_round_trip_reference_format(amd):
    """torch.save(model.state_dict()) / load_state_dict as at train_vqvae.py:173-182, 205-206."""
    dev = torch.device("cuda:0")
    cfg = O.TINY
    kw = dict(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
              embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m = amd.VQVAE(**kw)
    m.load_state_dict(O.make_state(cfg, 77))
    m.to(dev)
    tr = amd.Stage1Trainer(m, lr=3e-4, sched="cycle", n_iter=100)
    img = O.make_images(2, 32, 77).to(dev)
    tr.step(img)
    buf = io.BytesIO()
    torch.save(m.state_dict(), buf)
    buf.seek(0)
    sd = torch.load(buf, map_location="cpu", weights_only=True)
    assert list(sd.keys()) == list(O.state_spec(cfg).keys())
    m2 = amd.VQVAE(**kw)
    m2.load_state_dict(sd)
    m2.to(dev).eval()
    m.eval()
    with torch.no_grad():
        a, _ = m(img)
        b, _ = m2(img)
    close(a, b, rtol=0, atol=0)
    # the same checkpoint drives the CPU oracle (i.e. the reference layout) to the same output
    ref, _, _, _ = O.vqvae_forward({k: v.clone() for k, v in sd.items()}, cfg, img.cpu(), training=False)
    close(a, ref)


def _dx_close_up_to_relu_flips(dx, dx_ref, x, st, what, rtol=5e-4, atol=5e-5):
    """dx must match the oracle element-wise EXCEPT inside the receptive field of a mid-channel pre-activation
    h = conv3x3(relu(x)) + b1 (vqvae.py:86-87) that sits within fp32 summation error of zero: there the inner ReLU's
    mask (vqvae.py:88) may legitimately differ between two valid accumulation orders, which moves the 3x3x128
    input-gradient patch around that pixel by O(|dh|).  Anything else -- e.g. a tile-edge indexing slip, which
    would look similar in a plain allclose -- fails.  The exception must also stay rare, or the test means nothing."""
    bad = (dx - dx_ref).abs() > atol + rtol * dx_ref.abs()
    if not bool(bad.any()):
        return
    a = F.relu(x).double()
    w1, b1 = st["b.conv.1.weight"].double(), st["b.conv.1.bias"].double()
    hpre = F.conv2d(a, w1, b1, padding=1)                                   # [N,32,H,W] oracle pre-activation
    mag = F.conv2d(a.abs(), w1.abs(), b1.abs(), padding=1)                  # sum of |terms|: its rounding scale
    near = (hpre.abs() < 1e-5 * mag).any(1, keepdim=True)                   # [N,1,H,W]
    assert int(near.sum()) <= max(3, near.numel() // 20000), f"{what}: {int(near.sum())} near-zero pre-activations"
    field = F.max_pool2d(near.double(), 3, stride=1, padding=1) > 0         # 3x3 footprint of conv3x3's data gradient
    stray = bad & ~field.expand_as(bad)
    assert not bool(stray.any()), (f"{what}: {int(stray.sum())} of {int(bad.sum())} mismatching elements lie outside "
                                   f"the receptive field of a near-zero pre-activation (max err "
                                   f"{float((dx - dx_ref).abs()[stray].max()):.3e})")


@pytest.mark.parametrize("shape", [(1, 13, 21), (2, 8, 16), (3, 5, 40), (5, 64, 64)])
def test_fused_resblock_forward_backward(amd, shape):
    """One-launch ResBlock (csrc/vq2_resblock.hip; 128/32 channels) vs the oracle's vqvae.py:85-94:
    ragged tiles, channel-slice input/output, trailing ReLU."""
    from vqvae2_amd import ops
    dev = torch.device("cuda:0")
    n, h, w = shape
    blk = amd.ResBlock(128, 32)
    # seeded weights that differ from shape to shape (the default init would draw from torch's global RNG)
    tag = f"rb{n}x{h}x{w}"
    with torch.no_grad():
        blk.conv[1].weight.copy_(t(rng.normal(12, tag + ".w1", (32, 128, 3, 3))) * 0.03)
        blk.conv[1].bias.copy_(t(rng.normal(12, tag + ".b1", (32,))) * 0.1)
        blk.conv[3].weight.copy_(t(rng.normal(12, tag + ".w2", (128, 32, 1, 1))) * 0.15)
        blk.conv[3].bias.copy_(t(rng.normal(12, tag + ".b2", (128,))) * 0.1)
    blk.to(dev)
    assert ops.lib.vq2_resblock_supported(128, 32) == 1 and ops.lib.vq2_resblock_supported(32, 8) == 0
    st = {"b.conv.1.weight": blk.conv[1].weight.detach().cpu(), "b.conv.1.bias": blk.conv[1].bias.detach().cpu(),
          "b.conv.3.weight": blk.conv[3].weight.detach().cpu(), "b.conv.3.bias": blk.conv[3].bias.detach().cpu()}
    x_cpu = t(rng.normal(11, f"rb.x{shape}", (n, 128, h, w)))
    g_cpu = t(rng.normal(11, f"rb.g{shape}", (n, 128, h, w)))
    for relu_out in (False, True):
        for sliced in (False, True):
            wide = torch.zeros((n, h, w, 192), device=dev)
            xs = wide[..., 64:] if sliced else torch.empty((n, h, w, 128), device=dev)
            xs.copy_(x_cpu.permute(0, 2, 3, 1))
            xs.requires_grad_(True) if not sliced else None
            xin = xs if not sliced else xs.detach().requires_grad_(True)
            out_buf = torch.zeros((n, h, w, 160), device=dev)[..., :128] if sliced else None
            y = blk.nhwc(xin, relu_out=relu_out, out=out_buf)
            xr = x_cpu.clone().requires_grad_(True)
            stp = {k: v.clone().requires_grad_(True) for k, v in st.items()}
            ref = O.resblock(stp, "b", xr)
            if relu_out:
                ref = F.relu(ref)
            close(y.permute(0, 3, 1, 2), ref, what=f"y relu_out={relu_out} sliced={sliced}")
            for p in blk.parameters():
                p.grad = None
            y.backward(g_cpu.permute(0, 2, 3, 1).to(dev))
            ref.backward(g_cpu)
            _dx_close_up_to_relu_flips(xin.grad.permute(0, 3, 1, 2).cpu(), xr.grad, x_cpu, st,
                                       f"dx relu_out={relu_out} sliced={sliced}")
            for name, p in (("b.conv.1.weight", blk.conv[1].weight), ("b.conv.1.bias", blk.conv[1].bias),
                            ("b.conv.3.weight", blk.conv[3].weight), ("b.conv.3.bias", blk.conv[3].bias)):
                gref = stp[name].grad
                close(p.grad, gref, rtol=1e-3, atol=2e-5 * float(gref.abs().max()) + 1e-6, what=name)
    # the unfused composition (two conv launches) stays available and agrees
    ops.RESBLOCK_FUSED[0] = False
    try:
        y2 = blk.nhwc(xin.detach(), relu_out=True)
    finally:
        ops.RESBLOCK_FUSED[0] = True
    close(y2, blk.nhwc(xin.detach(), relu_out=True), rtol=1e-5, atol=1e-5)


def test_vq_stats_shapes_patterns_and_determinism(amd):
    """vq2_vq_stats (deterministic EMA statistics) driven directly with synthetic index patterns: chunk / block
    boundaries (63..1025 rows), one code for every row, one row per code, sorted and random assignments, a channel
    slice as input, D from 4 to 256 -- exact counts, sums against index_add, bit-identical on a second call."""
    import ctypes as C
    from vqvae2_amd import ops
    lib = ops.lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for M, D, K in [(1, 4, 4), (63, 16, 36), (64, 64, 512), (65, 64, 512), (1023, 8, 12), (1024, 64, 64), (1025, 32, 512),
                    (5000, 256, 100), (4099, 128, 8192), (70000, 64, 16)]:
        patterns = {
            "same": torch.full((M,), K - 1, dtype=torch.int64),
            "round": torch.arange(M, dtype=torch.int64) % K,
            "sorted": (torch.arange(M, dtype=torch.int64) * K) // M,
            "random": torch.randint(0, K, (M,), generator=g),
            "skewed": torch.minimum(torch.randint(0, K, (M,), generator=g), torch.randint(0, K, (M,), generator=g)) // 3,
        }
        wide = torch.randn(M, D + 8, generator=g)
        for name, idx in patterns.items():
            for sliced in (False, True):
                xh = wide[:, 4:4 + D] if sliced else wide[:, :D].contiguous()
                xd = wide.to(dev)[:, 4:4 + D] if sliced else xh.to(dev)
                ld = D + 8 if sliced else D
                idx_d = idx.to(dev)
                outs = []
                for _ in range(2):
                    stats = torch.full((K + K * D,), float("nan"), device=dev)      # every element must be written
                    nbytes = lib.vq2_vq_stats_workspace_bytes(M, D, K)
                    ws = torch.empty(nbytes // 4, device=dev, dtype=torch.int32)
                    ops.check(lib.vq2_vq_stats(C.c_void_p(xd.data_ptr()), ld, C.c_void_p(idx_d.data_ptr()), M, D, K,
                                               C.c_void_p(stats.data_ptr()), C.c_void_p(stats[K:].data_ptr()),
                                               C.c_void_p(ws.data_ptr()), nbytes, None), "vq_stats")
                    torch.cuda.synchronize()
                    outs.append(stats.cpu())
                assert torch.equal(outs[0], outs[1]), (M, D, K, name, sliced)
                counts = torch.bincount(idx, minlength=K).float()
                sums = torch.zeros(K, D, dtype=torch.float64).index_add_(0, idx, xh.double())
                assert torch.equal(outs[0][:K], counts), (M, D, K, name)
                np.testing.assert_allclose(outs[0][K:].reshape(K, D).numpy(), sums.numpy(), rtol=2e-5,
                                           atol=2e-5 * float(xh.abs().max()) * max(1.0, float(counts.max()) ** 0.5),
                                           err_msg=f"{(M, D, K, name, sliced)}")


def test_general_kernels_with_every_fast_path_switched_off():
    """The general kernels that the specialised ones replaced -- conv_gemm_kernel (no buffer-descriptor addressing),
    wgrad_kernel, the four-phase GEMM form of the sub-pixel layers, the two-launch ResBlock -- still back every shape the
    fast paths decline (k > 5, tensors beyond 2^29 elements).  Run the ragged-shape and fused-flag cases through them in
    a child process (the library reads its switches once per process)."""
    env = dict(os.environ, VQ2_FAST="0", VQ2_WFAST="0", VQ2_SUBPIX="0", VQ2_C4="0", VQ2_RB_FUSED="0", VQ2_WSWAP="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "ragged_conv_shapes or channel_slices"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_direct_forms_with_every_winograd_path_switched_off():
    """The direct implicit-GEMM / sub-pixel / fused-ResBlock kernels that the Winograd-domain forms replaced on rows of whole
    64-pixel segments still back every other shape (and every tensor beyond 1 GiB).  Run the model-level parity cases
    through them in a child process (the library reads its switches once per process)."""
    env = dict(os.environ, VQ2_WINO="0", VQ2_WINO_K4="0", VQ2_WINO_SP="0", VQ2_WWINO="0", VQ2_WWINO_K4="0", VQ2_WWINO_SW="0",
               VQ2_RB_WINO="0")
    parity = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py")
    r = subprocess.run([sys.executable, "-m", "pytest", parity, "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "full256 or 128px or tiny_vqvae_dropin"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]

