#!/bin/bash
# Interleaved A/B of scripts/microbench.py cases between libvq2.so (A) and libvq2_<tag>.so (B): bash scripts/ab_micro.sh <tag> "<cases>" [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; CASES=$2; ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  echo "== A (round $r)"; python3 $ROOT/scripts/microbench.py $CASES 2>&1 | grep -v amdgpu.ids
  echo "== B=$TAG (round $r)"; VQ2_LIB=$ROOT/vq-vae-2-pytorch_amd/libvq2_$TAG.so python3 $ROOT/scripts/microbench.py $CASES 2>&1 | grep -v amdgpu.ids
done
