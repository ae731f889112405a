"""Idle time of the main stream in one train step, from a `rocprofv3 --kernel-trace` CSV (scripts/trace_step.sh).

For every step (delimited by the step's first kernel, the NCHW->NHWC4 conversion of the image batch):
  wall            first kernel start -> next step's first kernel start
  main_busy       sum of main-stream kernel durations
  main_gaps       sum of (start[i+1] - end[i]) over consecutive main-stream kernels (idle between dependent launches)
  side_busy       sum of side-stream kernel durations (weight gradients of the 32x32-resolution layers, EMA statistics,
                  early slab reduction)
and the distribution of the gaps (how many, median, the largest with the kernels on either side).

    python scripts/stream_gaps.py gpurun_out/r3a_kernel_trace.csv [out.json]
"""
import csv
import json
import re
import statistics
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"vq2::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = [dict(name=short(r["Kernel_Name"]), stream=int(r["Stream_Id"]), t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"]))
          for r in rows]
    ks.sort(key=lambda k: k["t0"])
    main_stream = max(set(k["stream"] for k in ks), key=lambda s: sum(1 for k in ks if k["stream"] == s))
    starts = [i for i, k in enumerate(ks) if k["stream"] == main_stream and "nchw_to_nhwc" in k["name"]]
    steps = []
    for a, b in zip(starts[:-1], starts[1:]):
        seg = ks[a:b]
        m = [k for k in seg if k["stream"] == main_stream]
        s = [k for k in seg if k["stream"] != main_stream]
        gaps = [(m[i + 1]["t0"] - m[i]["t1"], m[i]["name"], m[i + 1]["name"]) for i in range(len(m) - 1)]
        gaps.append((ks[b]["t0"] - m[-1]["t1"], m[-1]["name"], ks[b]["name"]))
        steps.append({"wall_us": (ks[b]["t0"] - seg[0]["t0"]) / 1e3, "main_kernels": len(m), "side_kernels": len(s),
                      "main_busy_us": sum(k["t1"] - k["t0"] for k in m) / 1e3,
                      "side_busy_us": sum(k["t1"] - k["t0"] for k in s) / 1e3,
                      "main_gaps_us": sum(max(g[0], 0) for g in gaps) / 1e3,
                      "overlapped_us": -sum(min(g[0], 0) for g in gaps) / 1e3,
                      "gaps": gaps})
    steps = steps[len(steps) // 3:]          # skip the warm-up steps
    med = lambda key: round(statistics.median(s[key] for s in steps), 1)
    allg = sorted((g for s in steps for g in s["gaps"]), key=lambda g: -g[0])
    pos = [g[0] / 1e3 for g in allg if g[0] > 0]
    by_pair = {}
    for g in allg:
        d = by_pair.setdefault((g[1], g[2]), [])
        d.append(g[0] / 1e3)
    top = sorted(((statistics.median(v), len(v) // max(len(steps), 1), k) for k, v in by_pair.items()), reverse=True)[:12]
    out = {"steps_analysed": len(steps), "wall_us": med("wall_us"), "main_kernels": med("main_kernels"),
           "side_kernels": med("side_kernels"), "main_busy_us": med("main_busy_us"), "main_gaps_us": med("main_gaps_us"),
           "side_busy_us": med("side_busy_us"),
           "gap_frac_of_step": round(med("main_gaps_us") / med("wall_us"), 4),
           "gap_median_us": round(statistics.median(pos), 2), "gap_mean_us": round(sum(pos) / len(pos), 2),
           "largest_gaps_median_us": [{"us": round(t, 1), "per_step": n, "after": a, "before": b} for t, n, (a, b) in top]}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
