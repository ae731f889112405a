"""List the ATen ops (fill/copy/...) that a Stage1Trainer step still launches besides libvq2 kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import vqvae2_amd
from oracle import vqvae_oracle as O

dev = torch.device("cuda:0")
m = vqvae2_amd.VQVAE()
m.load_state_dict(O.make_state(O.DEFAULT, 1234))
m.to(dev)
tr = vqvae2_amd.Stage1Trainer(m, lr=3e-4)
img = O.make_images(8, 256, 1234).to(dev)
for _ in range(3):
    tr.step(img)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(img)
torch.cuda.synchronize()
from collections import Counter
c = Counter(ev.name for ev in prof.events())
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print("EV", v, k)
