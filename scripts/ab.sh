#!/bin/bash
# Interleaved A/B of two library builds in one process group: bash scripts/ab.sh <tagB> <script> [rounds]
#   A = vq-vae-2-pytorch_amd/libvq2.so, B = vq-vae-2-pytorch_amd/libvq2_<tagB>.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; SCRIPT=$2; ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  echo "== A (round $r)"; python3 $ROOT/$SCRIPT 2>&1 | grep -v amdgpu.ids
  echo "== B=$TAG (round $r)"; VQ2_LIB=$ROOT/vq-vae-2-pytorch_amd/libvq2_$TAG.so python3 $ROOT/$SCRIPT 2>&1 | grep -v amdgpu.ids
done
