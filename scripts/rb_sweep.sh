#!/bin/bash
# A/B of the fused ResBlock kernels' de-phased start: one process per setting (the library reads the env once).
#   bash scripts/rb_sweep.sh "0 8000 16000 24000 32000 48000"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for d in ${1:-0 16000 32000}; do
  echo "== dephase $d"
  VQ2_RB_DEPHASE_BWD=$d VQ2_RB_DEPHASE_FWD=$d python3 $ROOT/scripts/rb_occupancy.py 2>&1 | grep -E "N= 32 32x32|N= 32 64x64|N= 64 64x64"
done
