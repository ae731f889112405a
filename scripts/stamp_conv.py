"""Diagnostic: where a chunk iteration of the 128x128x32 conv tile spends its cycles (s_memtime)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd
from vqvae2_amd import ops
from vqvae2_amd.ops import ConvSpec
lib = vqvae2_amd._lib.lib
spec = ConvSpec(False, 128, 128, 3, 1, 1)
x = torch.randn(32, 64, 64, 128, device="cuda"); w = torch.randn(128, 128, 3, 3, device="cuda") * .05
b = torch.zeros(128, device="cuda")
for _ in range(5): ops.conv_forward(spec, x, w, b, 0)
buf = torch.zeros(16, dtype=torch.int64, device="cuda")
lib.vq2_debug_set_stamps(C.c_void_p(buf.data_ptr()))
for _ in range(3): ops.conv_forward(spec, x, w, b, 0)
torch.cuda.synchronize()
lib.vq2_debug_set_stamps(None)
t = buf.cpu().view(4, 4)
print("per wave: load-issue, mfma, store, barrier cycles (sum over 36 chunks; s_memtime @100MHz? or shader clk)")
for wv in range(4):
    r = t[wv].tolist(); tot = sum(r)
    print(wv, r, "per-chunk:", [round(v / 36) for v in r], "total/chunk", round(tot / 36))
