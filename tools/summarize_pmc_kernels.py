"""Per-kernel means of rocprofv3 --pmc passes (each pass in its own directory: <prefix>*/ ... _counter_collection.csv).
   python tools/summarize_pmc_kernels.py <dir prefix> <out.json> <kernel substring> [<kernel substring> ...]
For every kernel whose name contains one of the substrings: launches, mean duration in the PMC passes and the mean of
every counter collected; derived figures where their inputs are present (documented in the output)."""
import collections
import csv
import glob
import json
import sys


def main():
    prefix, out_path, wanted = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    meta = {}
    for f in sorted(glob.glob(f"{prefix}*/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = next((w for w in wanted if w in name), None)
            if key is None:
                continue
            key = f'{key} grid={r.get("Grid_Size", "?")} wg={r.get("Workgroup_Size", "?")}'
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
            meta[key] = {"vgpr": r.get("VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                         "lds_block_bytes": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size")}
    out = {}
    for key, counters in agg.items():
        c = {k: sum(v) / len(v) for k, v in counters.items()}
        t = sum(dur[key]) / len(dur[key])
        d = {"launches_seen": max(len(v) for v in counters.values()), "mean_seconds_in_pmc_passes": t, "resources": meta[key],
             "counters_mean_per_launch": c, "derived": {}}
        g = c.get("GRBM_GUI_ACTIVE")
        if g:
            cyc = g / 8.0                                           # summed over the 8 XCDs
            d["derived"]["gpu_cycles"] = cyc
            d["derived"]["effective_clock_GHz"] = cyc / t / 1e9
            if "SQ_ACTIVE_INST_VALU" in c:
                d["derived"]["valu_issue_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
            if "SQ_BUSY_CYCLES" in c:
                d["derived"]["sq_busy_frac"] = c["SQ_BUSY_CYCLES"] / g / 4 if c["SQ_BUSY_CYCLES"] > g else c["SQ_BUSY_CYCLES"] / g
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                d["derived"]["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
            if "SQ_WAVE_CYCLES" in c:
                d["derived"]["mean_waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc if c["SQ_WAVE_CYCLES"] else 0
        if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c and c["SQ_WAVES"]:
            d["derived"]["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            d["derived"]["wave_cycles_waiting_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rd, wr = c.get("FETCH_SIZE", 0) * 1024 * 2, c.get("WRITE_SIZE", 0) * 1024        # gfx950: FETCH_SIZE x2
            d["derived"]["hbm_bytes"] = {"read_FETCH_SIZE_x2": rd, "write_WRITE_SIZE": wr, "total": rd + wr,
                                         "GB_per_s": (rd + wr) / t / 1e9}
        out[key] = d
    json.dump(out, open(out_path, "w"), indent=1)
    for key, d in out.items():
        print(key, f'{d["mean_seconds_in_pmc_passes"] * 1e6:.1f} us', json.dumps(d["derived"]))


if __name__ == "__main__":
    main()
