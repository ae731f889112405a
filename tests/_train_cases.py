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


def run_dropin_case(amd, wrap_ddp):
    """The reference's own loop (train_vqvae.py:83-91, 166-171): stock nn.MSELoss + torch.optim.Adam, the model
    optionally wrapped in nn.parallel.DistributedDataParallel exactly as the reference wraps it."""
    cfg, size, batch, steps, seed = CASES["tiny"]
    m = amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                  embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m.load_state_dict(O.make_state(cfg, seed))
    m.cuda()
    model = m
    if wrap_ddp:
        # find_unused_parameters: the fork's dead `dec_ir` (vqvae.py:203-210) never receives a gradient, so plain DDP
        # stops at the second iteration ("Expected to have finished reduction ...") -- for the reference's module
        # exactly as for this one
        model = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], output_device=0,
                                                          find_unused_parameters=True)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    crit = torch.nn.MSELoss()
    losses = []
    for s in range(steps):
        img = O.make_images(batch, size, seed + s).cuda()
        opt.zero_grad()
        out, latent = model(img)
        loss = crit(out, img) + 0.25 * latent.mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, losses
