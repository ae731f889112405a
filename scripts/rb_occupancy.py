"""How much of the 32x32-resolution ResBlock gap is 'one workgroup per CU'?  Same kernels, same tile, the batch
doubled so that 512 workgroups (two per CU) are resident instead of 256: if the TFLOP/s barely move, a half-height
tile (which doubles the workgroups at batch 32 but also the weight-slice staging per pixel) cannot help either."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqvae2_amd

dev = torch.device("cuda:0")
lib = vqvae2_amd._lib.lib
blk = vqvae2_amd.ResBlock(128, 32).to(dev)
for n, hw in ((32, 32), (64, 32), (128, 32), (8, 64), (16, 64), (32, 64), (64, 64)):
    x = torch.randn(n, hw, hw, 128, device=dev)
    g = torch.randn(n, hw, hw, 128, device=dev)

    def run():
        xi = x.detach().requires_grad_(True)
        y = blk.nhwc(xi, relu_out=False)
        y.backward(g)
    import time
    t0 = time.time()
    while time.time() - t0 < 1.0:       # the clock needs ~1 s of load to settle (scripts/clock_probe.py)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    lib.vq2_prof_enable(1)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    lib.vq2_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.vq2_prof_report(buf, len(buf))
    row = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms, fl, by = line.split()
        if name.startswith("resblock"):
            row[name.split("|")[0]] = (float(ms) * 1e3 / int(cnt), float(fl) / float(ms) / 1e9)
    wgs = n * (hw // 8) * (hw // 16)
    # (workgroup count of the direct 8 x 16 tile; resblock_fwd_wino uses 4 x 64 tiles: half as many)
    print(f"N={n:3d} {hw}x{hw} workgroups={wgs:5d} ({wgs / 256:.1f}/CU): " +
          "  ".join(f"{k} {v[0]:6.1f} us {v[1]:6.1f} TF" for k, v in sorted(row.items())), flush=True)
