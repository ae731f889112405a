"""Diagnostic: phase timeline (s_memtime) of two workgroups of the fused ResBlock backward kernel."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd
lib = vqvae2_amd._lib.lib
dev = torch.device("cuda:0")
blk = vqvae2_amd.ResBlock(128, 32).to(dev)
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(32, HW, HW, 128, device=dev)
g = torch.randn(32, HW, HW, 128, device=dev)
def run():
    xi = x.detach().requires_grad_(True)
    blk.nhwc(xi).backward(g)
import time
t0 = time.time()
while time.time() - t0 < 2.0:      # settle the clock first (scripts/clock_probe.py)
    for _ in range(20): run()
    torch.cuda.synchronize()
buf = torch.zeros(256, dtype=torch.int64, device=dev)
lib.vq2_debug_set_rb_stamps(C.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.vq2_debug_set_rb_stamps(None)
allt = buf.cpu()[:128].view(2, 2, 4, 8)
fine = buf.cpu()[128:160].view(2, 4, 4)
if int(fine.sum()):
    print('forward kernel, slices 1..7 of wave 0 (VQ2_RB_FINE build): store / issue / mfma / barrier cycles per slice:', [round(v / 7) for v in fine[0, 0].tolist()])
for kern, names in ((1, ["start->slice0 staged", "stage 1 (8 slices)", "r write", "stage 2", "epilogue"]),
                    (0, ["start->A0 staged", "phase A", "dh write + tap0", "phase B", "epilogue"])):
  t = allt[kern]
  print("forward kernel" if kern else "backward kernel")
  for b in range(2):
    for w in range(4):
        r = t[b, w].tolist()
        if r[5] == 0: continue
        print(f"wg{b} wave{w}: " + "  ".join(f"{names[i]}={r[i + 1] - r[i]}" for i in range(5)) + f"  total={r[5] - r[0]}  clock={(r[5] - r[0]) / max(r[7] - r[6], 1) * 0.1:.2f} GHz")
