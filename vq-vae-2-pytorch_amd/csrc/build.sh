#!/bin/bash
# Build libvq2.so (gfx950 only) in-tree next to the Python package.
set -euo pipefail
trap 'echo "build.sh: FAILED (see the compiler messages above or csrc/_obj/*.res)" >&2' ERR
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${VQ2_OUT:-$HERE/../libvq2.so}"     # VQ2_OUT + VQ2_OBJ + VQ2_EXTRA_FLAGS: A/B variant builds (scripts/build_variant.sh)
OBJ="${VQ2_OBJ:-$HERE/_obj}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function"
mkdir -p "$OBJ"
pids=()
for f in vq2_conv vq2_wino vq2_rbwino vq2_wgrad vq2_vq vq2_elem vq2_resblock vq2_norm; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/vq2_common.h" -nt "$OBJ/$f.o" ] || [ "$HERE/vq2_conv.h" -nt "$OBJ/$f.o" ] || [ "$HERE/vq2_rbwino.h" -nt "$OBJ/$f.o" ] || [ "$HERE/../../include/vq2.h" -nt "$OBJ/$f.o" ]; then
    # the compiler's per-kernel resource report (VGPRs, spills, scratch, occupancy) is kept next to the object:
    # tests/test_host_cpu.py::test_hot_kernels_do_not_spill reads it (a spilled register in a conv tile cost 3 % of
    # the step in round 2 and no functional test could see it)
    ( $HIPCC $FLAGS -Rpass-analysis=kernel-resource-usage -c "$HERE/$f.hip" -o "$OBJ/$f.o" ${VQ2_EXTRA_FLAGS:-} \
        2> "$OBJ/$f.res"; rc=$?; grep -E -A4 "(error|warning):" "$OBJ/$f.res" >&2 || true; exit $rc ) &
    pids+=($!)
  fi
done
$HIPCC $FLAGS -x hip -c "$HERE/vq2_core.cpp" -o "$OBJ/vq2_core.o" &
pids+=($!)
$HIPCC $FLAGS -x hip -I/opt/rocm/include -c "$HERE/vq2_comm.cpp" -o "$OBJ/vq2_comm.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC -shared -fPIC --offload-arch=gfx950 "$OBJ"/*.o -ldl -o "$OUT"
echo "built $OUT"
