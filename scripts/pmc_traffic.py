"""profiles/pmc_traffic.json from the PMC summaries of a round (scripts/collect_profiles.sh): per workload, the memory-side
bytes per launch of the dominant conv tile that bench.py reports as roofline.traffic.

    python scripts/pmc_traffic.py r03 [dir with <round>_<workload>_pmc_summary.json, default profiles/]

FETCH_SIZE is already doubled by scripts/pmc_summary.py (gfx950 tallies 128-byte requests at 64 bytes,
MI355X_MICROARCH.md HBM section); WRITE_SIZE is exact for 16-byte streaming stores."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
# bench.py's kernel-family label -> the template instances behind it in the rocprofv3 kernel names
FAMILIES = {"conv_gemm<128x128x16>": r"conv_gemm_fast_kernel<2, 2, 2, 2, 16,",
            "conv_gemm<128x128x32>": r"conv_gemm_fast_kernel<2, 2, 2, 2, 32,",
            "conv_wino3<2x64,nt2>": r"(vq2::)?(wino::)?wino3_kernel<32, 2,",
            "conv_wino_k4s2<2x64>": r"(vq2::)?(wino::)?wino_k4s2_kernel<",
            "conv_wino_subpixel<4x64>": r"(vq2::)?(wino::)?wino_subpixel_kernel<"}
BATCH = {"c2": 32, "c4": 32, "c5": 8}
out = {}
for w in ("c2", "c4", "c5"):
    f = os.path.join(src, f"{rnd}_{w}_pmc_summary.json")
    if not os.path.exists(f):
        continue
    pm = json.load(open(f))
    for label, pat in FAMILIES.items():
        inst = {k: v for k, v in pm.items() if re.match(pat, k) and "fetch_MB_per_launch" in v and "write_MB_per_launch" in v}
        n = sum(v["launches"] for v in inst.values())
        if not n:
            continue
        fetch = sum(v["fetch_MB_per_launch"] * v["launches"] for v in inst.values()) / n * 1e6
        write = sum(v["write_MB_per_launch"] * v["launches"] for v in inst.values()) / n * 1e6
        hit = sum(v.get("l2_hit", 0) * v["launches"] for v in inst.values()) / n
        out.setdefault(w, {})[label] = {
            "batch": BATCH[w], "bytes_per_launch": round(fetch + write), "fetch_bytes_per_launch": round(fetch),
            "write_bytes_per_launch": round(write), "l2_hit": round(hit, 3), "instances": sorted(inst),
            "source": f"profiles/{rnd}_{w}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                      f"bench.py --workload {w} --steps 4, scripts/collect_profiles.sh; FETCH_SIZE x2 per MI355X_MICROARCH.md HBM "
                      f"section; launch-weighted over {len(inst)} template instance(s))"}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print({w: {k: v["bytes_per_launch"] for k, v in d.items()} for w, d in out.items()})
