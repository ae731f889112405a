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


def _run(world, rank, imgs, steps):
    import vqvae2_amd
    cfg = O.TINY
    m = vqvae2_amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                         embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    # every rank starts from a DIFFERENT state (what per-process RNG init gives a real launch): the trainer must
    # bring all of them to rank 0's, like DistributedDataParallel's constructor (train_vqvae.py:166-171)
    m.load_state_dict(O.make_state(cfg, 1234 + 100 * rank))
    m.cuda()
    tr = vqvae2_amd.Stage1Trainer(m, lr=3e-4)
    assert tr.world == world
    for _ in range(steps):
        out = tr.step(imgs.cuda())
    torch.cuda.synchronize()
    if world > 1:   # the decoder-side gradient bucket went out from the backward hook (overlap path), every step
        assert tr.split_off is not None and tr.early_buckets == steps
    return {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, float(out["loss"])


def _worker(rank, world, port, out, wgrad_stream):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["VQ2_WGRAD_STREAM"] = wgrad_stream
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = O.make_images(4, 32, 1234)
    sd, loss = _run(world, rank, full[rank * 2:(rank + 1) * 2].contiguous(), 2)
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
