"""The minimal-filtering identities behind csrc/vq2_wino.hip, vq2_rbwino.hip and wgrad_fast_kernel<..., WINO>, restated in fp64
torch on the CPU and checked against torch's own convolutions (no GPU, no libvq2): the transforms the kernels apply to
inputs (V), taps (U), gradients (E) and outputs / slabs are exactly the ones below ("Minimal filtering").
A wrong sign or column offset here would be a wrong kernel there; the GPU parity tests then compare the kernels themselves."""
import torch
import torch.nn.functional as F

torch.manual_seed(5)
D = torch.float64


def _pairs(x):
    """x [N, C, H, W] (W even) -> padded columns d0..d3 of every output pair: each [N, C, H, W/2]."""
    xp = F.pad(x, (1, 1, 0, 0))                  # columns -1 .. W
    w2 = x.shape[-1] // 2
    return [xp[..., c::2][..., :w2] for c in range(4)]      # column 2t - 1 + c


def test_f23_rows_forward_3x3():
    n, ci, co, h, w = 2, 5, 7, 6, 8
    x = torch.randn(n, ci, h, w, dtype=D)
    g = torch.randn(co, ci, 3, 3, dtype=D)
    ref = F.conv2d(x, g, padding=1)
    xr = F.pad(x, (0, 0, 1, 1))                   # rows -1 .. H
    y = torch.zeros(n, co, h, w, dtype=D)
    m = [torch.zeros(n, co, h, w // 2, dtype=D) for _ in range(4)]
    for kh in range(3):
        d0, d1, d2, d3 = _pairs(xr[:, :, kh:kh + h])
        v = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)                                        # wino3_kernel: fv[]
        g0, g1, g2 = g[:, :, kh, 0], g[:, :, kh, 1], g[:, :, kh, 2]
        u = (g0, g0 + g1 + g2, g0 - g1 + g2, g2)                                        # store_b (1/2 moved to the output)
        for i in range(4):
            m[i] += torch.einsum("nchw,oc->nohw", v[i], u[i])
    y[..., 0::2] = m[0] + 0.5 * (m[1] + m[2])                                           # store_pairs
    y[..., 1::2] = 0.5 * (m[1] - m[2]) - m[3]
    assert torch.allclose(y, ref, rtol=0, atol=1e-12)


def test_f22_by_column_parity_k4s2_forward():
    n, ci, co, h, w = 2, 3, 4, 8, 16               # input h x w, output h/2 x w/2, pairs of outputs: w/4
    x = torch.randn(n, ci, h, w, dtype=D)
    g = torch.randn(co, ci, 4, 4, dtype=D)
    ref = F.conv2d(x, g, stride=2, padding=1)
    xp = F.pad(x, (1, 3, 1, 1))                    # column index c <-> xp[..., c + 1]
    ho, wo = h // 2, w // 2
    a = [torch.zeros(n, co, ho, wo // 2, dtype=D) for _ in range(3)]
    for kh in range(4):
        rows = xp[:, :, kh:kh + 2 * ho:2]          # input rows 2 ho + kh - 1
        for par in range(2):                       # 0: odd columns (taps 0, 2), 1: even columns (taps 1, 3)
            pc = -1 if par == 0 else 0
            d = [rows[..., (pc + 2 * j + 1)::4][..., :wo // 2] for j in range(3)]       # columns 4t + pc + 2j
            ga, gb = g[:, :, kh, par], g[:, :, kh, par + 2]
            v = (d[0] - d[1], d[1], d[1] - d[2])                                        # wino_k4s2_kernel: fv[]
            u = (ga, ga + gb, gb)
            for i in range(3):
                a[i] += torch.einsum("nchw,oc->nohw", v[i], u[i])
    y = torch.zeros_like(ref)
    y[..., 0::2] = a[0] + a[1]
    y[..., 1::2] = a[1] - a[2]
    assert torch.allclose(y, ref, rtol=0, atol=1e-12)


def test_f22_per_phase_conv_transpose_forward():
    n, ci, co, h, w = 2, 3, 4, 4, 8
    x = torch.randn(n, ci, h, w, dtype=D)
    g = torch.randn(ci, co, 4, 4, dtype=D)         # IOHW
    ref = F.conv_transpose2d(x, g, stride=2, padding=1)
    xp = F.pad(x, (1, 2, 1, 1))                    # column c <-> xp[..., c + 1], row r <-> xp[:, :, r + 1]
    y = torch.zeros_like(ref)
    for ph in range(2):
        for pw in range(2):
            a = [torch.zeros(n, co, h, w // 2, dtype=D) for _ in range(3)]
            for ar in range(2):                    # kernel row a of the phase's 2x2 filter: input row h + a - 1 + ph
                rows = xp[:, :, ar + ph:ar + ph + h]
                d = [rows[..., (pw + j)::2][..., :w // 2] for j in range(3)]            # columns 2t - 1 + pw + j
                # tap of input offset (a, b) for output phase (ph, pw): kh = 3 - 2a - ph, kw = 3 - 2b - pw
                g0 = g[:, :, 3 - 2 * ar - ph, 3 - pw]
                g1 = g[:, :, 3 - 2 * ar - ph, 1 - pw]
                v = (d[0] - d[1], d[1], d[1] - d[2])                                    # wino_subpixel_kernel: fv[]
                u = (g0, g0 + g1, g1)
                for i in range(3):
                    a[i] += torch.einsum("nchw,co->nohw", v[i], u[i])
            y[:, :, ph::2, pw::4] = a[0] + a[1]                                         # outputs 2(2t) + pw and 2(2t+1) + pw
            y[:, :, ph::2, pw + 2::4] = a[1] - a[2]
    assert torch.allclose(y, ref, rtol=0, atol=1e-12)


def _wgrad_ref(x, dy, stride, pad, k):
    w = torch.zeros(dy.shape[1], x.shape[1], k, k, dtype=D, requires_grad=True)
    F.conv2d(x, w, stride=stride, padding=pad).backward(dy)
    return w.grad


def test_f23_weight_gradient_over_column_pairs():
    n, ci, co, h, w = 2, 4, 3, 5, 8
    x = torch.randn(n, ci, h, w, dtype=D)
    dy = torch.randn(n, co, h, w, dtype=D)
    ref = _wgrad_ref(x, dy, 1, 1, 3)
    xr = F.pad(x, (0, 0, 1, 1))
    g0, g1 = dy[..., 0::2], dy[..., 1::2]
    e = (g0, g0 + g1, g0 - g1, g1)                                                       # WINO 1: E (the last one with V' = -V3)
    dw = torch.zeros_like(ref)
    for kh in range(3):
        d0, d1, d2, d3 = _pairs(xr[:, :, kh:kh + h])
        v = (d0 - d2, d1 + d2, d2 - d1, d3 - d1)
        s = [torch.einsum("nohw,nchw->oc", e[i], v[i]) for i in range(4)]
        half = 0.5 * (s[1] + s[2])                                                       # wino_reduce_unit, mode 1
        dw[:, :, kh, 0] = s[0] + half
        dw[:, :, kh, 1] = 0.5 * (s[1] - s[2])
        dw[:, :, kh, 2] = half + s[3]
    assert torch.allclose(dw, ref, rtol=0, atol=1e-11)


def test_f22_weight_gradient_k4s2_by_parity():
    n, ci, co, h, w = 2, 3, 4, 8, 16
    x = torch.randn(n, ci, h, w, dtype=D)
    ho, wo = h // 2, w // 2
    dy = torch.randn(n, co, ho, wo, dtype=D)
    ref = _wgrad_ref(x, dy, 2, 1, 4)
    xp = F.pad(x, (1, 3, 1, 1))
    g0, g1 = dy[..., 0::2], dy[..., 1::2]
    e = (g0, g0 + g1, g1)                                                                # WINO 2
    dw = torch.zeros_like(ref)
    for kh in range(4):
        rows = xp[:, :, kh:kh + 2 * ho:2]
        for par in range(2):
            pc = -1 if par == 0 else 0
            d = [rows[..., (pc + 2 * j + 1)::4][..., :wo // 2] for j in range(3)]
            v = (d[0] - d[1], d[1], d[1] - d[2])
            s = [torch.einsum("nohw,nchw->oc", e[i], v[i]) for i in range(3)]
            dw[:, :, kh, par] = s[0] + s[1]                                              # wino_reduce_unit, mode 2
            dw[:, :, kh, par + 2] = s[1] - s[2]
    assert torch.allclose(dw, ref, rtol=0, atol=1e-11)
    # the conv-transpose bias gradient from the middle-product tiles: d1 of both parities on input rows 2 ho (kh 1) and
    # 2 ho + 1 (kh 2), plus the spare load on columns 4t (even parity) / 4t + 3 (odd): every pixel of the gathered tensor once
    total = torch.zeros(ci, dtype=D)
    for kh in (1, 2):
        rows = xp[:, :, kh:kh + 2 * ho:2]
        for par in range(2):
            pc = -1 if par == 0 else 0
            d1 = rows[..., (pc + 2 + 1)::4][..., :wo // 2]
            spare = rows[..., ((0 if par else 3) + 1)::4][..., :wo // 2]
            total += (d1 + spare).sum((0, 2, 3))
    assert torch.allclose(total, x.sum((0, 2, 3)), rtol=0, atol=1e-11)


def test_f23_weight_gradient_with_exchanged_roles():
    n, ci, co, h, w = 2, 6, 3, 5, 8                 # the ResBlock 3x3: many input channels, few output channels
    x = torch.randn(n, ci, h, w, dtype=D)
    dy = torch.randn(n, co, h, w, dtype=D)
    ref = _wgrad_ref(x, dy, 1, 1, 3)
    d0, d1, d2, d3 = _pairs(x)                       # V on the STREAMED operand x (row r), with column checks (zero padding)
    v = (d0 - d2, d1 + d2, d2 - d1, d3 - d1)
    dyr = F.pad(dy, (0, 0, 1, 1))                    # E on the gathered dy at row r - kh + 1
    dw = torch.zeros_like(ref)
    for kh in range(3):
        rows = dyr[:, :, 2 - kh:2 - kh + h]          # dy row r - kh + 1  <->  padded index r - kh + 2
        g0, g1 = rows[..., 0::2], rows[..., 1::2]
        e = (g0, g0 + g1, g0 - g1, g1)
        s = [torch.einsum("nohw,nchw->oc", e[i], v[i]) for i in range(4)]
        half = 0.5 * (s[1] + s[2])                   # wino_reduce_unit, mode 3 (destination transposed there)
        dw[:, :, kh, 0] = s[0] + half
        dw[:, :, kh, 1] = 0.5 * (s[1] - s[2])
        dw[:, :, kh, 2] = half + s[3]
    assert torch.allclose(dw, ref, rtol=0, atol=1e-11)
