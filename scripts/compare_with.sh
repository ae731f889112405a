#!/bin/bash
# Same-box A/B of the WHOLE tree against an earlier commit (e.g. the previous round's final one):
#     bash scripts/compare_with.sh <git-ref> [bench.py arguments]        # here, in the build container
#     gpurun -- 'bash scripts/compare_with.sh --run [bench.py arguments]' # then on the GPU box
# Step 1 checks the ref out into ./_ref (git-ignored, travels with the gpurun snapshot) and builds its library;
# step 2 alternates `_ref/bench.py` and `bench.py` in ONE call, so both trees see the same box, clocks and neighbours.
# Round 2 lost most of a day to a 3 % regression that every flag-vs-flag A/B inside the new tree shared; this is the
# comparison that found it.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"
if [ "${1:-}" != "--run" ]; then
  REF="${1:?usage: compare_with.sh <git-ref> | --run}"
  git worktree remove --force _ref 2>/dev/null || true
  git worktree add -f _ref "$REF" -q
  grep -qx "_ref/" .git/info/exclude || echo "_ref/" >> .git/info/exclude
  (cd _ref && bash vq-vae-2-pytorch_amd/csrc/build.sh | tail -1)
  echo "now: gpurun -- 'bash scripts/compare_with.sh --run'"
  exit 0
fi
shift
ARGS="${*:---steps 30 --warmup 10 --no-cpu-baseline --no-prof}"
for i in 1 2 3; do
  (cd _ref && python bench.py $ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/ref   /')
  python bench.py $ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/head  /'
done
