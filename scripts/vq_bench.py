"""Time vq_fwd alone (train: with EMA statistics; eval: without) on bench-sized inputs."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqvae2_amd

dev = torch.device("cuda:0")
lib = vqvae2_amd._lib.lib
for k in (512, 8192):
    q = vqvae2_amd.Quantize(64, k).to(dev)
    for hw in (64, 32):
        for spread in (1.0, 0.05):   # 0.05: nearly collapsed codebook usage (synthetic-data regime)
            x = torch.randn(32, hw, hw, 64, device=dev) * spread
            for mode in ("train", "eval"):
                q.train(mode == "train")
                for _ in range(3):
                    q(x)
                torch.cuda.synchronize()
                lib.vq2_prof_enable(1)
                for _ in range(10):
                    q(x)
                torch.cuda.synchronize()
                lib.vq2_prof_enable(0)
                buf = ctypes.create_string_buffer(1 << 16)
                lib.vq2_prof_report(buf, len(buf))
                for line in buf.value.decode().splitlines():
                    name, cnt, ms, fl, by = line.split()
                    if name.startswith("vq_"):
                        print(f"K={k} {hw}x{hw} spread={spread} {mode} {name}: {float(ms) * 1e3 / int(cnt):7.1f} us  {float(fl) / float(ms) / 1e9:6.1f} TF", flush=True)
