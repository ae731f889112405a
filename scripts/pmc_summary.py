"""Fold rocprofv3 --pmc counter CSVs (separate passes) into per-kernel HBM-side bytes per launch.

    python scripts/pmc_summary.py OUT.json PASS_DIR [PASS_DIR ...]

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it
is doubled (MI355X_MICROARCH.md, HBM section).  Kernel names are shortened to the template instance."""
import csv, glob, json, os, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("void vq2::", "").replace("vq2::", "")
            name = name.split("(")[0]
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[name][row["Counter_Name"]] += 1
res = {}
for k, c in acc.items():
    n = max(cnt[k].values())
    e = {"launches": n}
    if "FETCH_SIZE" in c:
        e["fetch_MB_per_launch"] = round(2.0 * c["FETCH_SIZE"] * 1024 / cnt[k]["FETCH_SIZE"] / 1e6, 1)
    if "WRITE_SIZE" in c:
        e["write_MB_per_launch"] = round(c["WRITE_SIZE"] * 1024 / cnt[k]["WRITE_SIZE"] / 1e6, 1)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        e["l2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the 1,024 SIMD matrix pipes; GRBM_GUI_ACTIVE sums the
        # active cycles of the 8 XCDs (MI355X_MICROARCH.md, DVFS note): busy / (active / 8 * 1024) = pipe utilisation
        e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 3)
    for extra in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
        if extra in c:
            e[extra + "_per_launch"] = round(c[extra] / cnt[k][extra])
    res[k] = e
res = dict(sorted(res.items(), key=lambda kv: -(kv[1].get("fetch_MB_per_launch", 0) + kv[1].get("write_MB_per_launch", 0)) * kv[1]["launches"]))
json.dump(res, open(out, "w"), indent=1)
print(f"{len(res)} kernels -> {out}")
