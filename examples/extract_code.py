"""Encode-only export on the MI355X path (/root/reference/extract_code.py:14-33, 59-68).

model.eval() -> model.encode(img) -> (id_t [B,H/8,W/8], id_b [B,H/4,W/4]) int64.  The reference pickles
CodeRow(top, bottom, filename) rows keyed str(index) plus a 'length' key into LMDB (dataset.py:11,
36-51); lmdb is not installed here, so the same rows go to one .npz (top, bottom, filename arrays) --
format pinned from the reference's code only.

    python examples/extract_code.py --ckpt checkpoint/vqvae_001.pt --path images_dir --name codes.npz
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--ckpt", type=str, required=True)
    ap.add_argument("--name", type=str, required=True)
    ap.add_argument("--path", type=str, required=True, help="directory of .npy batches [N,3,H,W] float32")
    args = ap.parse_args()
    device = torch.device("cuda:0")
    model = vqvae2_amd.VQVAE()
    model.load_state_dict(torch.load(args.ckpt, map_location="cpu", weights_only=True))
    model = model.to(device).eval()                           # extract_code.py:59-62
    tops, bottoms, names = [], [], []
    with torch.no_grad():
        for f in sorted(glob.glob(os.path.join(args.path, "*.npy"))):
            img = torch.from_numpy(np.load(f)).float().to(device)
            _, _, _, id_t, id_b = model.encode(img)           # extract_code.py:23
            tops.append(id_t.cpu().numpy())
            bottoms.append(id_b.cpu().numpy())
            names += [f"{os.path.basename(f)}:{i}" for i in range(img.shape[0])]
    np.savez_compressed(args.name, top=np.concatenate(tops), bottom=np.concatenate(bottoms),
                        filename=np.array(names), length=len(names))
    print(f"wrote {len(names)} code rows to {args.name}")


if __name__ == "__main__":
    main()
