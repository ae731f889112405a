"""Per-kernel time of one train step from a rocprofv3 --kernel-trace CSV (scripts/trace_step.sh): median over the
analysed steps of the summed duration per (kernel, grid, stream).   python scripts/trace_table.py <csv> [out.json]"""
import collections
import csv
import json
import re
import statistics
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"vq2::", "", name)
    return re.sub(r"\(.*$", "", name)[:80]


rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((short(r["Kernel_Name"]), int(r["Stream_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
              int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows), key=lambda k: k[2])
starts = [i for i, k in enumerate(ks) if "nchw_to_nhwc" in k[0]]
steps = list(zip(starts[:-1], starts[1:]))
steps = steps[len(steps) // 3:]
main = max({k[1] for k in ks}, key=lambda s: sum(1 for k in ks if k[1] == s))
per = collections.defaultdict(list)
counts = {}
for a, b in steps:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k in ks[a:b]:
        e = agg[(k[0], "main" if k[1] == main else "side", k[4])]
        e[0] += 1
        e[1] += (k[3] - k[2]) / 1e3
    for key, (n, us) in agg.items():
        per[key].append(us)
        counts[key] = n
table = sorted(((statistics.median(v), counts[k], k) for k, v in per.items()), reverse=True)
tot_main = sum(t for t, n, k in table if k[1] == "main")
tot_side = sum(t for t, n, k in table if k[1] == "side")
print(f"steps analysed {len(steps)}; kernel time per step: main stream {tot_main:.0f} us, side streams {tot_side:.0f} us")
for t, n, (name, stream, grid) in table:
    print(f"{t:8.1f} us  n={n:2d} {stream:4s} grid={grid:5d}  {name}")
if len(sys.argv) > 2:
    json.dump({"steps_analysed": len(steps), "main_stream_us": round(tot_main, 1), "side_streams_us": round(tot_side, 1),
               "kernels": [{"us_per_step": round(t, 1), "launches": n, "stream": k[1], "workgroups": k[2], "kernel": k[0]}
                           for t, n, k in table]}, open(sys.argv[2], "w"), indent=1)
