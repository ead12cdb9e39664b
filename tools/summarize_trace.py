"""Per-kernel durations from a rocprofv3 --kernel-trace CSV, grouped by (kernel, grid): count, mean, min, p50, max
in microseconds. The launches of one kernel at different grids (the local and the remote force block) stay
separate.   python tools/summarize_trace.py <..._kernel_trace.csv> [out.json]"""
import csv
import json
import re
import statistics
import sys
from collections import defaultdict


def main():
    rows = defaultdict(list)
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("kernel_name")
            m = re.search(r"(\w+)(<[^(]*>)?\(", name)          # function name + template arguments, no namespaces
            name = ((m.group(1) + (m.group(2) or "")) if m else name)[:72]
            grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            wg = r.get("Workgroup_Size_X", "?")
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            rows[(name, grid, wg)].append(dur)
    out = []
    for (name, grid, wg), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        d2 = sorted(d)
        steady = d2[: max(1, int(len(d2) * 0.9))]                # drop the slowest 10 % (clock ramp, first launches)
        out.append({"kernel": name, "grid_threads": grid, "workgroup": wg, "launches": len(d),
                    "mean_us": statistics.fmean(d), "mean_us_fastest_90pct": statistics.fmean(steady),
                    "min_us": d2[0], "p50_us": d2[len(d2) // 2], "max_us": d2[-1], "total_ms": sum(d) / 1e3})
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    for o in out[:25]:
        print(f'{o["kernel"][:56]:56s} grid {o["grid_threads"]:>14s} n={o["launches"]:5d} mean {o["mean_us"]:9.2f} '
              f'p50 {o["p50_us"]:9.2f} min {o["min_us"]:9.2f} us')


if __name__ == "__main__":
    main()
