#!/bin/bash
# Build libvq2.so (gfx950 only) in-tree next to the Python package.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libvq2.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function"
mkdir -p "$HERE/_obj"
pids=()
for f in vq2_conv vq2_wgrad vq2_vq vq2_elem vq2_resblock vq2_norm; do
  if [ ! -f "$HERE/_obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/_obj/$f.o" ] || [ "$HERE/vq2_common.h" -nt "$HERE/_obj/$f.o" ] || [ "$HERE/../../include/vq2.h" -nt "$HERE/_obj/$f.o" ]; then
    # the compiler's per-kernel resource report (VGPRs, spills, scratch, occupancy) is kept next to the object:
    # tests/test_host_cpu.py::test_hot_kernels_do_not_spill reads it (a spilled register in a conv tile cost 3 % of
    # the step in round 2 and no functional test could see it)
    ( $HIPCC $FLAGS -Rpass-analysis=kernel-resource-usage -c "$HERE/$f.hip" -o "$HERE/_obj/$f.o" ${VQ2_EXTRA_FLAGS:-} \
        2> "$HERE/_obj/$f.res"; rc=$?; grep -E -A4 "(error|warning):" "$HERE/_obj/$f.res" >&2 || true; exit $rc ) &
    pids+=($!)
  fi
done
$HIPCC $FLAGS -x hip -c "$HERE/vq2_core.cpp" -o "$HERE/_obj/vq2_core.o" &
pids+=($!)
$HIPCC $FLAGS -x hip -I/opt/rocm/include -c "$HERE/vq2_comm.cpp" -o "$HERE/_obj/vq2_comm.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC -shared -fPIC --offload-arch=gfx950 "$HERE"/_obj/*.o -ldl -o "$OUT"
echo "built $OUT"
