"""Digest of the model state after two train steps (configs[1] geometry, batch 8): run under two library builds
(VQ2_LIB=...) to check that a kernel change left every bit where it was."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqvae2_amd
from oracle import vqvae_oracle as O
m = vqvae2_amd.VQVAE()
m.load_state_dict(O.make_state(O.DEFAULT, 77))
m.cuda()
tr = vqvae2_amd.Stage1Trainer(m, lr=3e-4)
for s in range(2):
    tr.step(O.make_images(int(os.environ.get("BATCH", "8")), 256, 77 + s).cuda())
torch.cuda.synchronize()
h = hashlib.sha256()
for k, v in m.state_dict().items():
    h.update(v.detach().cpu().numpy().tobytes())
print("state digest", h.hexdigest()[:16])
