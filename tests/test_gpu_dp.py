"""GPU: Stage1Trainer's data-parallel path (initial broadcast from rank 0, flat arena, packed all-reduces of
gradients + EMA statistics in two buckets, deferred EMA update, Adam with 1/world scaling) with 2 ranks sharing the
one GPU of the test box over gloo, against the single-process step on the concatenated batch.  (RCCL refuses two
ranks on one device; tests/test_gpu_rccl.py runs the same trainer path over RCCL at world size 1.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vqvae_oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, rank, imgs, steps, warm=False):
    import vqvae2_amd
    cfg = O.TINY
    m = vqvae2_amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                         embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    # every rank starts from a DIFFERENT state (what per-process RNG init gives a real launch): the trainer must
    # bring all of them to rank 0's, like DistributedDataParallel's constructor (train_vqvae.py:166-171)
    m.load_state_dict(O.make_state(cfg, 1234 + 100 * rank))
    m.cuda()
    if warm:
        # a drop-in training forward BEFORE the trainer exists: every Quantize now caches the prepared form (embedT,
        # ||e||^2) of ITS OWN post-update codebook; the trainer's initial broadcast overwrites `embed` through .data /
        # raw pointers and must drop that cache, or ranks != 0 keep searching their old codebook
        m.train()
        with torch.no_grad():
            m(imgs.cuda())
        assert m.quantize_t._prepared() is not None
    tr = vqvae2_amd.Stage1Trainer(m, lr=3e-4)
    assert tr.world == world
    for _ in range(steps):
        out = tr.step(imgs.cuda())
    torch.cuda.synchronize()
    if world > 1:   # the tail AND the middle bucket went out from backward hooks (overlap path), every step
        assert [b[0] for b in tr.buckets] == ["tail", "middle", "head"] and tr.early_buckets == 2 * steps
        lo_hi = sorted((lo, hi) for _, lo, hi, _ in tr.buckets)       # the three slices tile the gradient buffer
        assert lo_hi[0][0] == 0 and lo_hi[-1][1] == tr.arena.flat_g.numel()
        assert all(a[1] == b[0] for a, b in zip(lo_hi[:-1], lo_hi[1:]))
    return {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, float(out["loss"])


def _worker(rank, world, port, out, wgrad_stream, warm=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["VQ2_WGRAD_STREAM"] = wgrad_stream
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = O.make_images(4, 32, 1234)
    sd, loss = _run(world, rank, full[rank * 2:(rank + 1) * 2].contiguous(), 2, warm)
    np.savez(out + f".rank{rank}.npz", **sd)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wgrad_stream", ["0", "1"])
def test_two_rank_trainer_equals_single_rank_on_full_batch(tmp_path, wgrad_stream):
    out = str(tmp_path / "dp_gpu")
    mp.spawn(_worker, args=(2, _free_port(), out, wgrad_stream), nprocs=2, join=True)
    r0, r1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    ref, _ = _run(1, 0, O.make_images(4, 32, 1234), 2)
    for k, v in ref.items():
        assert np.array_equal(r0[k], r1[k]), f"{k}: replicas diverged"      # dec_ir and the codebooks included
        if not k.startswith("dec_ir."):
            np.testing.assert_allclose(r0[k], v, rtol=1e-3, atol=2e-5, err_msg=k)


def test_initial_broadcast_drops_the_prepared_codebook_cache(tmp_path):
    """Ranks that already ran a drop-in forward hold a prepared-codebook cache of their OWN codebook; after the
    trainer's initial broadcast they must search rank 0's (replicas bit-identical, and equal to one rank that started
    from rank 0's state and saw the whole batch)."""
    out = str(tmp_path / "dp_warm")
    mp.spawn(_worker, args=(2, _free_port(), out, "1", True), nprocs=2, join=True)
    r0, r1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), f"{k}: replicas diverged (stale prepared codebook on rank 1?)"


def _ddp_worker(rank, world, port, out):
    """The reference's own data-parallel recipe (train_vqvae.py:166-171, 83-91) on the drop-in module at world 2:
    nn.parallel.DistributedDataParallel + stock Adam / MSELoss, so Quantize.forward's in-forward
    dist_fn.all_reduce of the EMA statistics (vqvae.py:58-59, distributed.py:64-72) runs on DEVICE tensors."""
    import vqvae2_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd, losses = _ddp_run(vqvae2_amd, world, rank)
    np.savez(out + f".rank{rank}.npz", **sd)
    dist.barrier()
    dist.destroy_process_group()


def _ddp_run(amd, world, rank, steps=2, batch=4):
    cfg = O.TINY
    m = amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                  embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m.load_state_dict(O.make_state(cfg, 77 + 100 * rank))       # DDP's constructor broadcasts rank 0's state
    m.cuda()
    model = m
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], output_device=0,
                                                          find_unused_parameters=True)   # dead dec_ir, vqvae.py:203-210
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    crit = torch.nn.MSELoss()
    per = batch // world
    losses = []
    for s in range(steps):
        img = O.make_images(batch, 32, 300 + s)[rank * per:(rank + 1) * per].contiguous().cuda()
        opt.zero_grad()
        dec, latent = model(img)
        loss = crit(dec, img) + 0.25 * latent.mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return {k.replace("module.", "", 1): v.detach().cpu().numpy() for k, v in model.state_dict().items()}, losses


def test_two_rank_ddp_wrapped_dropin_module_equals_single_rank():
    import tempfile
    import vqvae2_amd
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "ddp")
        mp.spawn(_ddp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
        r0, r1 = dict(np.load(out + ".rank0.npz")), dict(np.load(out + ".rank1.npz"))
    ref, _ = _ddp_run(vqvae2_amd, 1, 0)
    for k, v in ref.items():
        assert np.array_equal(r0[k], r1[k]), f"{k}: replicas diverged"
        if not k.startswith("dec_ir."):
            # the EMA statistics are exact sums of both ranks' rows; gradients are means of two half-batch means
            np.testing.assert_allclose(r0[k], v, rtol=1e-3, atol=2e-5, err_msg=k)
