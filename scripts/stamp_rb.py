"""Diagnostic: phase timeline (s_memtime) of two workgroups of the fused ResBlock backward kernel."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd
lib = vqvae2_amd._lib.lib
dev = torch.device("cuda:0")
blk = vqvae2_amd.ResBlock(128, 32).to(dev)
x = torch.randn(32, 64, 64, 128, device=dev)
g = torch.randn(32, 64, 64, 128, device=dev)
def run():
    xi = x.detach().requires_grad_(True)
    blk.nhwc(xi).backward(g)
for _ in range(5): run()
buf = torch.zeros(64, dtype=torch.int64, device=dev)
lib.vq2_debug_set_rb_stamps(C.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.vq2_debug_set_rb_stamps(None)
t = buf.cpu().view(2, 4, 8)
names = ["start->A0 staged", "phase A", "dh write + tap0", "phase B", "epilogue"]
for b in range(2):
    for w in range(4):
        r = t[b, w].tolist()
        print(f"wg{b} wave{w}: " + "  ".join(f"{names[i]}={r[i + 1] - r[i]}" for i in range(5)) + f"  total={r[5] - r[0]}  start={r[0] - int(t[0, 0, 0])}")
