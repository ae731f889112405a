"""Throughput of the inference paths SURVEY §8f ranks next (extract_code.py:14-33 encode, sample.py:97-100
decode_code), synthetic 256x256 input resident in HBM, eval mode, no autograd.  Prints one JSON line per path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqvae2_amd
from oracle import vqvae_oracle as O

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
m = vqvae2_amd.VQVAE()
m.load_state_dict(O.make_state(O.DEFAULT, 1234))
m.to(dev).eval()
img = O.make_images(B, 256, 1234).to(dev)
with torch.no_grad():
    _, _, _, id_t, id_b = m.encode(img)

def timed(fn, steps=30, warmup=10):
    with torch.no_grad():
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps

for name, fn, gflop in (("encode (image -> id_t, id_b)", lambda: m.encode(img), 2 * (1526.7 + 293.6 + 8.4 + 293.6 + 50.3 + 167.8) * 1e-3),
                        ("decode_code (id_t, id_b -> image)", lambda: m.decode_code(id_t, id_b), 2 * (67.1 + 1526.7) * 1e-3),
                        ("forward (reconstruction, eval)", lambda: m(img), 7.87)):
    dt = timed(fn)
    print(json.dumps({"path": name, "batch": B, "ms": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1),
                      "TFLOP/s": round(gflop * B / dt / 1e3, 1)}), flush=True)
