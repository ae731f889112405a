"""Stage-1 trainer on the MI355X path with the reference's command line.

Mirrors /root/reference/train_vqvae.py:209-237 (flags), :144-206 (main: model, optimizer, optional
CycleScheduler, --resume, checkpoint every 10 epochs as checkpoint/vqvae_{epoch:03d}.pt in the
reference's state_dict format) and :27-141 (per-step: recon MSE + 0.25 * latent, running mse
aggregated over ranks).  The re-ID parts of the fork are out of scope.  Data: a directory of .npy
image batches ([N,3,H,W] float32, already normalised) or, without --path, synthetic N(0,1) images
(there is no torchvision / dataset access in this environment).

    python examples/train_stage1.py --size 256 --batch_size 32 --epoch 1 --iters 50
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_stage1.py ...
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch
import torch.distributed as tdist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd  # noqa: E402
from vqvae2_amd import distributed as dist  # noqa: E402


def batches(args, rank, world, device):
    if args.path:
        files = sorted(glob.glob(os.path.join(args.path, "*.npy")))[rank::world]
        for f in files:
            arr = torch.from_numpy(np.load(f)).float()
            for i in range(0, arr.shape[0] - args.batch_size + 1, args.batch_size):
                yield arr[i:i + args.batch_size].to(device, non_blocking=True)
    else:
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        for _ in range(args.iters):
            yield torch.randn(args.batch_size, 3, args.size, args.size, generator=g).to(device)


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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        tdist.init_process_group("nccl", device_id=device)
    rank = dist.get_rank()

    model = vqvae2_amd.VQVAE().to(device)
    if args.resume:                                           # train_vqvae.py:173-182
        sd = torch.load(args.resume, map_location="cpu", weights_only=True)
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        model.load_state_dict(sd)
    n_iter = (args.iters if not args.path else max(1, len(glob.glob(os.path.join(args.path, "*.npy"))))) * args.epoch
    trainer = vqvae2_amd.Stage1Trainer(model, lr=args.lr, sched=args.sched, n_iter=n_iter)

    for epoch in range(args.epoch):
        mse_sum = torch.zeros(2, device=device)                # (sum of recon * batch, count)
        for i, img in enumerate(batches(args, rank, world, device)):
            out = trainer.step(img)
            mse_sum[0] += out["recon"] * img.shape[0]
            mse_sum[1] += img.shape[0]
            if i % 25 == 0:
                agg = mse_sum.clone()
                dist.all_reduce(agg)                          # replaces the pickled all_gather of train_vqvae.py:93-100
                if dist.is_primary():
                    lr = trainer.optimizer.param_groups[0]["lr"]
                    print(f"epoch: {epoch + 1}; it {i}; mse: {float(out['recon']):.5f}; "
                          f"latent: {float(out['latent']):.3f}; avg mse: {float(agg[0] / agg[1]):.5f}; lr: {lr:.5f}",
                          flush=True)
        if dist.is_primary() and (epoch % 10 == 0 or epoch == args.epoch - 1):   # train_vqvae.py:205-206
            os.makedirs(args.out, exist_ok=True)
            torch.save(model.state_dict(), os.path.join(args.out, f"vqvae_{str(epoch + 1).zfill(3)}.pt"))
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
