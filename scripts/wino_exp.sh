#!/bin/bash
# timing experiments of the Winograd conv (variant libraries built with -DVQ2_WINO_EXP=n): bash scripts/wino_exp.sh e1 e2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== base"; MB_REPEAT=1 python3 $ROOT/scripts/microbench.py c3_128_128 2>&1 | grep -v amdgpu.ids
for t in "$@"; do
  echo "== $t"; MB_REPEAT=1 VQ2_LIB=$ROOT/vq-vae-2-pytorch_amd/libvq2_$t.so python3 $ROOT/scripts/microbench.py c3_128_128 2>&1 | grep -v amdgpu.ids
done
