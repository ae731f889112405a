"""Do two half-batch chains on two streams beat one full-batch chain?  (the tails of dependent launches cannot overlap
on one stream; two independent chains can fill each other's tails.)  Chain = L dependent 3x3 128->128 convs at 64x64
(the dominant kernel), or conv -> ResBlock -> ResBlock -> conv (the decoder's 64x64 level, forward only).
usage: python scripts/two_chain_probe.py [conv|mixed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqvae2_amd
from vqvae2_amd import ops
from vqvae2_amd.ops import ConvSpec

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
L = 6
spec = ConvSpec(False, 128, 128, 3, 1, 1)
ws = [torch.randn(128, 128, 3, 3, device=dev) * 0.02 for _ in range(L)]
bs = [torch.zeros(128, device=dev) for _ in range(L)]
blk = vqvae2_amd.ResBlock(128, 32).to(dev)


def chain(x, outs):
    with torch.no_grad():
        for i in range(L):
            if kind == "mixed" and i % 3 != 0:
                x = blk.nhwc(x, relu_out=False)
            else:
                x = ops.conv_forward(spec, x, ws[i], bs[i], ops.VQ2_RELU_IN, out=outs[i])
    return x


def bench(fn, secs=2.0, iters=20):
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


N = 32
x = torch.randn(N, 64, 64, 128, device=dev)
outs = [torch.empty(N, 64, 64, 128, device=dev) for _ in range(L)]
PR = [int(v) for v in os.environ.get("PRIO", "0,0").split(",")]
s1, s2 = torch.cuda.Stream(priority=PR[0]), torch.cuda.Stream(priority=PR[1])


def one():
    chain(x, outs)


def halves(parts):
    def run():
        cur = torch.cuda.current_stream()
        strs = (s1, s2)
        n = N // parts
        for p in range(parts):
            s = strs[p % 2]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                chain(x[p * n:(p + 1) * n], [o[p * n:(p + 1) * n] for o in outs])
        for s in strs:
            cur.wait_stream(s)
    return run


def serial_halves():
    n = N // 2
    for p in range(2):
        chain(x[p * n:(p + 1) * n], [o[p * n:(p + 1) * n] for o in outs])


for rnd in range(2):
    print(f"{kind} L={L}: one stream N=32 {bench(one):8.1f} us | two streams 2 x N=16 {bench(halves(2)):8.1f} us | "
          f"two streams 4 x N=8 {bench(halves(4)):8.1f} us | one stream 2 x N=16 {bench(serial_halves):8.1f} us", flush=True)
