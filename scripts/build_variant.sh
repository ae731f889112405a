#!/bin/bash
# Build a second libvq2 from the working tree with extra compiler flags, for A/B runs in ONE gpurun call:
#   bash scripts/build_variant.sh old "-DVQ2_RB_WRITE_AFTER=0"   ->  vq-vae-2-pytorch_amd/libvq2_old.so
#   VQ2_LIB=$PWD/vq-vae-2-pytorch_amd/libvq2_old.so python scripts/rb_occupancy.py
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
TAG=$1; shift
VQ2_OUT="$ROOT/vq-vae-2-pytorch_amd/libvq2_$TAG.so" VQ2_OBJ="/tmp/vq2_obj_$TAG" VQ2_EXTRA_FLAGS="$*" bash "$ROOT/vq-vae-2-pytorch_amd/csrc/build.sh"
