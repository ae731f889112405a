"""Shared by the RCCL child process and its parent test: the same seeded training run on either side."""
import torch

from oracle import vqvae_oracle as O

CASES = {
    # name: (config, image size, batch, steps, seed)
    "tiny": (O.TINY, 32, 4, 3, 61),
    "default64": (O.DEFAULT, 64, 4, 2, 62),
}


def run_case(amd, case, seed_offset=0):
    cfg, size, batch, steps, seed = CASES[case]
    m = amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                  embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m.load_state_dict(O.make_state(cfg, seed + seed_offset))
    m.cuda()
    tr = amd.Stage1Trainer(m, lr=3e-4)
    losses = []
    for s in range(steps):
        out = tr.step(O.make_images(batch, size, seed + s).cuda())
        losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    return {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, losses, tr
