"""Stage-1 trainer on the MI355X path with the reference's command line.

Mirrors /root/reference/train_vqvae.py:209-237 (flags), :144-206 (main: model, optimizer, optional
CycleScheduler, --resume, checkpoint every 10 epochs as checkpoint/vqvae_{epoch:03d}.pt in the
reference's state_dict format) and :27-141 (per-step: recon MSE + 0.25 * latent, running mse
aggregated over ranks).  The re-ID parts of the fork are out of scope.  Data: a directory of .npy
image batches ([N,3,H,W] float32, already normalised) or, without --path, synthetic N(0,1) images
(there is no torchvision / dataset access in this environment).

Every rank runs the SAME number of steps: the batches found under --path are dealt round-robin and the
remainder that would give some ranks one step more is dropped (what DistributedSampler's equal shards do for
the reference), and the CycleScheduler's n_iter is that per-rank count x epochs (train_vqvae.py:189-195).

    python examples/train_stage1.py --size 256 --batch_size 32 --epoch 1 --iters 50
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_stage1.py ...
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd  # noqa: E402
from vqvae2_amd import distributed as dist  # noqa: E402


def plan_batches(args, rank, world):
    """-> list of (file, first row) this rank trains on per epoch (None = synthetic), equal length on every rank."""
    if not args.path:
        return [None] * args.iters
    every = []
    for f in sorted(glob.glob(os.path.join(args.path, "*.npy"))):
        rows = np.load(f, mmap_mode="r").shape[0]
        every += [(f, i) for i in range(0, rows - args.batch_size + 1, args.batch_size)]
    common = len(every) // world
    if common == 0:
        raise SystemExit(f"{len(every)} batches of {args.batch_size} under {args.path}: fewer than the {world} ranks")
    return every[rank::world][:common]


def load_batch(item, args, gen, device):
    if item is None:
        return torch.randn(args.batch_size, 3, args.size, args.size, generator=gen).to(device)
    f, i = item
    return torch.from_numpy(np.ascontiguousarray(np.load(f, mmap_mode="r")[i:i + args.batch_size])).float().to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n_gpu", type=int, default=1)            # kept for CLI parity; world size comes from the launcher
    ap.add_argument("--dist_url", default="env://")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--epoch", type=int, default=560)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--sched", type=str)
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--resume", "-r", default="", type=str)
    ap.add_argument("--path", type=str, default="")
    ap.add_argument("--iters", type=int, default=100, help="synthetic batches per epoch when --path is not given")
    ap.add_argument("--out", default="checkpoint")
    args = ap.parse_args()

    rank, local_rank, world = dist.bringup("nccl")             # launch.py:52-92 (RCCL group + device binding)
    device = torch.device("cuda", local_rank)

    # every rank may build its model from its own RNG state: the trainer broadcasts rank 0's (DDP, train_vqvae.py:166-171)
    model = vqvae2_amd.VQVAE().to(device)
    plan = plan_batches(args, rank, world)
    trainer = vqvae2_amd.Stage1Trainer(model, lr=args.lr, sched=args.sched, n_iter=len(plan) * args.epoch)
    first_epoch = 0
    if args.resume:                                            # train_vqvae.py:173-182
        sd = torch.load(args.resume, map_location=device, weights_only=True)
        trainer.load_state_dict(sd)                            # bare model state_dict or a trainer checkpoint
        first_epoch = int(sd.get("epoch", 0)) if "model" in sd else 0
        if dist.is_primary():
            print(f"==> loaded checkpoint {args.resume} (epoch {first_epoch})")

    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    for epoch in range(first_epoch, args.epoch):
        mse_sum = torch.zeros(2, device=device)                # (sum of recon * batch, count)
        for i, item in enumerate(plan):
            img = load_batch(item, args, gen, device)
            out = trainer.step(img)
            mse_sum[0] += out["recon"] * img.shape[0]
            mse_sum[1] += img.shape[0]
            if i % 25 == 0:
                agg = mse_sum.clone()
                dist.all_reduce(agg)                           # replaces the pickled all_gather of train_vqvae.py:93-100
                if dist.is_primary():
                    lr = trainer.optimizer.param_groups[0]["lr"]
                    print(f"epoch: {epoch + 1}; it {i}; mse: {float(out['recon']):.5f}; "
                          f"latent: {float(out['latent']):.3f}; avg mse: {float(agg[0] / agg[1]):.5f}; lr: {lr:.5f}",
                          flush=True)
        if dist.is_primary() and (epoch % 10 == 0 or epoch == args.epoch - 1):   # train_vqvae.py:205-206
            os.makedirs(args.out, exist_ok=True)
            tag = str(epoch + 1).zfill(3)
            torch.save(model.state_dict(), os.path.join(args.out, f"vqvae_{tag}.pt"))       # the reference's file
            full = trainer.state_dict()
            full["epoch"] = epoch + 1
            torch.save(full, os.path.join(args.out, f"trainer_{tag}.pt"))                   # exact-resume extras
    dist.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
