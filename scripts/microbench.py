"""Per-layer microbenchmark of libvq2 conv kernels (HIP-event timing), for kernel iteration.
usage: python scripts/microbench.py [case ...]   cases: c3_128_128 c3_128_32 c1_32_128 c4s2_64_128 t_128_64 ...
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd  # noqa: E402
from vqvae2_amd import ops  # noqa: E402
from vqvae2_amd.ops import ConvSpec  # noqa: E402

CASES = {
    # name: (spec, N, H, W)
    "c3_128_128": (ConvSpec(False, 128, 128, 3, 1, 1), 32, 64, 64),
    "c3_128_32": (ConvSpec(False, 128, 32, 3, 1, 1), 32, 64, 64),
    "c1_32_128": (ConvSpec(False, 32, 128, 1, 1, 0), 32, 64, 64),
    "c4s2_64_128": (ConvSpec(False, 64, 128, 4, 2, 1), 32, 128, 128),
    "c4s2_3_64": (ConvSpec(False, 3, 64, 4, 2, 1), 32, 256, 256),
    "t_128_64": (ConvSpec(True, 128, 64, 4, 2, 1), 32, 64, 64),
    "t_64_3": (ConvSpec(True, 64, 3, 4, 2, 1), 32, 128, 128),
    "c1_192_64": (ConvSpec(False, 192, 64, 1, 1, 0), 32, 64, 64),
    "c3_32_128": (ConvSpec(False, 32, 128, 3, 1, 1), 32, 64, 64),   # shape of the ResBlock 3x3 data gradient
    # 32x32-resolution layers (enc_t / dec_t)
    "s_c3_64_128": (ConvSpec(False, 64, 128, 3, 1, 1), 32, 32, 32),
    "s_c3_128_32": (ConvSpec(False, 128, 32, 3, 1, 1), 32, 32, 32),
    "s_c4s2_128_64": (ConvSpec(False, 128, 64, 4, 2, 1), 32, 64, 64),
    "s_t_128_64": (ConvSpec(True, 128, 64, 4, 2, 1), 32, 32, 32),
    "s_t_64_64": (ConvSpec(True, 64, 64, 4, 2, 1), 32, 32, 32),
    "s_c1_128_64": (ConvSpec(False, 128, 64, 1, 1, 0), 32, 32, 32),
}


def timeit(fn, iters=30):
    for _ in range(15):   # let the clocks settle
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    names = sys.argv[1:] or list(CASES)
    dev = torch.device("cuda:0")
    for name in names:
        spec, n, h, w = CASES[name]
        x = torch.randn(n, h, w, spec.ci, device=dev)
        wshape = (spec.cin, spec.cout, spec.k, spec.k) if spec.transposed else (spec.cout, spec.cin, spec.k, spec.k)
        wt = torch.randn(wshape, device=dev) * 0.05
        b = torch.randn(spec.cout, device=dev)
        ho, wo = spec.out_hw(h, w)
        dy = torch.randn(n, ho, wo, spec.co, device=dev)
        macs = (n * h * w * 16 if spec.transposed else n * ho * wo * spec.k * spec.k) * spec.cin * spec.cout
        fl = 2.0 * macs
        relu = 0 if os.environ.get("MB_NORELU") else ops.VQ2_RELU_IN
        t_f = timeit(lambda: ops.conv_forward(spec, x, wt, b, relu))
        t_d = timeit(lambda: ops.conv_dgrad(spec, x.shape, dy, wt, mask=None if os.environ.get('MB_NOMASK') else x))
        t_w = timeit(lambda: ops.conv_wgrad(spec, x, dy, not os.environ.get('MB_NORELU'), wt, None if os.environ.get('MB_NOBIAS') else b))
        if os.environ.get("MB_REPEAT"):
            t_f = timeit(lambda: ops.conv_forward(spec, x, wt, b, relu))   # again, after the clocks have settled
        print(f"{name:14s} fwd {t_f:8.1f} us {fl / t_f / 1e6:6.1f} TF | dgrad {t_d:8.1f} us {fl / t_d / 1e6:6.1f} TF | "
              f"wgrad {t_w:8.1f} us {fl / t_w / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
