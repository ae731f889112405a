"""World-size-2 data-parallel equivalence on CPU (gloo): the packed SUM all-reduce design of
train.py -- gradients averaged (DDP semantics, train_vqvae.py:166-171), EMA statistics summed
(vqvae.py:58-59) -- reproduces the single-process step on the concatenated batch.  Exercises the
oracle's hooks (the algorithm) and vqvae2_amd.distributed (the mirrored helpers) under gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vqvae_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vqvae2_amd.distributed as d
    assert d.get_world_size() == world and d.get_rank() == rank and d.is_primary() == (rank == 0)
    t = torch.full((3,), float(rank + 1))
    assert d.all_reduce(t) is t and float(t[0]) == 3.0               # distributed.py:64-72
    assert d.all_gather({"r": rank}) == [{"r": 0}, {"r": 1}]         # distributed.py:75-107
    red = d.reduce_dict({"a": torch.tensor(float(rank))})
    if rank == 0:
        assert abs(float(red["a"]) - 0.5) < 1e-6
    d.synchronize()

    cfg = O.TINY
    st = O.make_state(cfg, 1234)
    adam = O.AdamState({k: v for k, v in st.items() if not O.is_buffer(k)})
    full = O.make_images(4, 32, 1234)
    img = full[rank * 2:(rank + 1) * 2]

    def ema_sum(x):            # vqvae.py:58-59
        dist.all_reduce(x)

    def grad_mean(g):          # DDP
        dist.all_reduce(g)
        g /= world

    for _ in range(2):
        r = O.train_step(st, cfg, img, adam, all_reduce=ema_sum, grad_all_reduce=grad_mean)
    if rank == 0:
        np.savez(out, **{k: v.numpy() for k, v in st.items()}, loss=r["loss"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_full_batch(tmp_path):
    out = str(tmp_path / "dp.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    cfg = O.TINY
    st = O.make_state(cfg, 1234)
    adam = O.AdamState({k: v for k, v in st.items() if not O.is_buffer(k)})
    full = O.make_images(4, 32, 1234)
    for _ in range(2):
        O.train_step(st, cfg, full, adam)
    for k, v in st.items():
        if k.startswith("dec_ir."):
            continue
        # identical in exact arithmetic (every cross-rank quantity is a SUM); fp32 reassociation only
        np.testing.assert_allclose(got[k], v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
