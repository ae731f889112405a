"""Encode-only export on the MI355X path (/root/reference/extract_code.py:14-33, 59-68).

model.eval() -> model.encode(img) -> (id_t [B,H/8,W/8], id_b [B,H/4,W/4]) int64, written as the reference's rows:
pickle.dumps(CodeRow(top, bottom, filename)) under str(index) keys plus a 'length' key (dataset.py:11, 36-51;
vqvae2_amd/codes.py).  The container is LMDB when `lmdb` is importable, else one sqlite3 file with the same keys
and the same value bytes.

Images: class sub-folders of image files like the reference's ImageFolder (resize, centre crop, [-1,1]
normalisation: extract_code.py:46-53) when PIL can open them, or .npy batches [N,3,H,W] float32 already normalised.

    python examples/extract_code.py --ckpt checkpoint/vqvae_001.pt --name codes.db images_dir
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd  # noqa: E402
from vqvae2_amd import codes  # noqa: E402

IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp")


def resize_crop_box(w, h, size):
    """((new_w, new_h), (left, top, right, bottom)) of transforms.Resize(size) followed by transforms.CenterCrop(size)
    (extract_code.py:48-51) with torchvision's own integer rules: the short side becomes `size`, the long side is
    TRUNCATED int(size * long / short); the crop offsets are int(round((dim - size) / 2.0))."""
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    return (nw, nh), (left, top, left + size, top + size)


def image_batches(path, size, batch=128):
    """ImageFolder order (sorted classes, sorted walk, sorted files); filename = immediate parent directory / file
    (dataset.py:14-22 takes the LAST directory component, which differs from the class for nested folders)."""
    from PIL import Image
    files = []
    for cls in sorted(d for d in os.listdir(path) if os.path.isdir(os.path.join(path, d))):
        for root, _, names in sorted(os.walk(os.path.join(path, cls), followlinks=True)):
            files += [(os.path.join(root, n), os.path.join(os.path.basename(root), n))
                      for n in sorted(names) if n.lower().endswith(IMG_EXT)]
    for i in range(0, len(files), batch):
        arrs = []
        for full, _ in files[i:i + batch]:
            im = Image.open(full).convert("RGB")
            new_size, box = resize_crop_box(*im.size, size)
            a = np.asarray(im.resize(new_size, Image.BILINEAR).crop(box), np.float32) / 255.0
            arrs.append((a.transpose(2, 0, 1) - 0.5) / 0.5)                          # ToTensor + Normalize(0.5, 0.5)
        yield torch.from_numpy(np.stack(arrs)), [name for _, name in files[i:i + batch]]


def npy_batches(path):
    for f in sorted(glob.glob(os.path.join(path, "*.npy"))):
        arr = np.load(f)
        yield torch.from_numpy(arr).float(), [f"{os.path.basename(f)}:{i}" for i in range(arr.shape[0])]


def extract(store, loader, model, device):
    """extract_code.py:14-33."""
    index = 0
    for img, filename in loader:
        img = img.to(device)
        _, _, _, id_t, id_b = model.encode(img)
        id_t = id_t.detach().cpu().numpy()
        id_b = id_b.detach().cpu().numpy()
        index = codes.write_code_rows(store, id_t, id_b, filename, start=index)
    store.put("length".encode("utf-8"), str(index).encode("utf-8"))
    return index


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--ckpt", type=str, required=True)
    ap.add_argument("--name", type=str, required=True)
    ap.add_argument("path", type=str)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    model = vqvae2_amd.VQVAE()
    sd = torch.load(args.ckpt, map_location="cpu", weights_only=True)
    model.load_state_dict({(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()})
    model = model.to(device).eval()                            # extract_code.py:59-62
    has_npy = bool(glob.glob(os.path.join(args.path, "*.npy")))
    loader = npy_batches(args.path) if has_npy else image_batches(args.path, args.size)
    with torch.no_grad(), codes.CodeStore(args.name, "w") as store:
        n = extract(store, loader, model, device)
    print(f"inserted: {n} rows into {args.name}")


if __name__ == "__main__":
    main()
