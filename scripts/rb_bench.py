"""Time the ResBlock forward/backward launches alone (HIP events on the launch stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqvae2_amd
from vqvae2_amd import ops

dev = torch.device("cuda:0")
blk = vqvae2_amd.ResBlock(128, 32).to(dev)
for hw in (64, 32):
    x = torch.randn(32, hw, hw, 128, device=dev)
    g = torch.randn(32, hw, hw, 128, device=dev)
    flops = 2.0 * 32 * hw * hw * (9 * 128 * 32 + 32 * 128)
    for mode in ("fwd", "fwd+bwd"):
        def run():
            xi = x.detach().requires_grad_(mode != "fwd")
            with torch.set_grad_enabled(mode != "fwd"):
                y = blk.nhwc(xi, relu_out=False)
            if mode != "fwd":
                y.backward(g)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 30
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        import ctypes
        lib = vqvae2_amd._lib.lib
        lib.vq2_prof_enable(1)
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        lib.vq2_prof_enable(0)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.vq2_prof_report(buf, len(buf))
        for line in buf.value.decode().splitlines():
            name, cnt, ms, fl, by = line.split()
            print(f"    {name:60s} {float(ms) * 1e3 / int(cnt):8.1f} us  {float(fl) / float(ms) / 1e9:7.1f} TF")
        print(f"{hw}x{hw} {mode}: {us:.1f} us  ({flops * (1 if mode == 'fwd' else 3) / us / 1e6:.1f} TF)", flush=True)
