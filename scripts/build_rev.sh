#!/bin/bash
# Build libvq2 from the kernel sources of a git revision, for same-box A/B against the working tree:
#   bash scripts/build_rev.sh HEAD~1 prev   ->  vq-vae-2-pytorch_amd/libvq2_prev.so     (then scripts/ab.sh prev bench.py ...)
# (timings of different boxes / different gpurun calls differ by 1-3 %: compare builds inside ONE call)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REV=$1; TAG=$2
T=/tmp/vq2_rev_$TAG
rm -rf $T && mkdir -p $T
git -C "$ROOT" archive "$REV" vq-vae-2-pytorch_amd/csrc include | tar -x -C $T
cp "$ROOT/vq-vae-2-pytorch_amd/csrc/build.sh" $T/vq-vae-2-pytorch_amd/csrc/build.sh
VQ2_OUT="$ROOT/vq-vae-2-pytorch_amd/libvq2_$TAG.so" VQ2_OBJ="$T/obj" bash $T/vq-vae-2-pytorch_amd/csrc/build.sh
