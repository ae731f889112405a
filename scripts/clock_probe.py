"""In-kernel clock of the dominant conv instantiation (the F(2,3) Winograd kernel, or with VQ2_WINO=0 the direct four-per-CU
128x128x16 tile; 3x3 128->128 at 64x64, batch 32):
lifetime of four workgroups in shader cycles (s_memtime) and in 10 ns ticks (s_memrealtime), after two seconds of
back-to-back launches (MI355X_MICROARCH.md, DVFS give-back item 6).   VQ2_CLOCKPROBE=1 python scripts/clock_probe.py"""
import ctypes as C, os, sys, time, torch
os.environ["VQ2_CLOCKPROBE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd
from vqvae2_amd import ops
from vqvae2_amd.ops import ConvSpec
lib = vqvae2_amd._lib.lib
spec = ConvSpec(False, 128, 128, 3, 1, 1)
x = torch.randn(32, 64, 64, 128, device="cuda"); w = torch.randn(128, 128, 3, 3, device="cuda") * .05
b = torch.zeros(128, device="cuda")
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50): ops.conv_forward(spec, x, w, b, 0)
    torch.cuda.synchronize()
buf = torch.zeros(16, dtype=torch.int64, device="cuda")
lib.vq2_debug_set_stamps(C.c_void_p(buf.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.conv_forward(spec, x, w, b, 0)
e1.record()
torch.cuda.synchronize()
lib.vq2_debug_set_stamps(None)
us = e0.elapsed_time(e1) * 1e3 / 20
fl = 2.0 * 32 * 64 * 64 * 9 * 128 * 128
print(f"launch {us:.1f} us = {fl / us / 1e6:.1f} TFLOP/s")
for slot, r in enumerate(buf.cpu().view(4, 4).tolist()):
    cyc, ticks, mfmas = r[0], r[1], r[2]
    if not ticks: continue
    clk = cyc / ticks * 0.1
    res = 4 if os.environ.get("VQ2_WINO") == "0" else 2     # resident waves per SIMD (workgroups per CU)
    print(f"workgroup {8 + 256 * slot}: {cyc} cycles in {ticks * 10} ns -> {clk:.2f} GHz; {mfmas} MFMAs per wave x 64 cycles = "
          f"{mfmas * 64} pipe cycles; with {res} waves per SIMD resident the pipe needs {res * mfmas * 64} cycles per round of tiles -> "
          f"utilisation {res * mfmas * 64 / cyc:.2f}; fp32 MFMA peak at this clock {256 * 4 * 64 * clk / 1e3:.0f} TFLOP/s")
