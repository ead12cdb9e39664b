"""Turns the rocprofv3 CSVs of one profiling session (gpurun_out/) into the committed summaries
under profiles/: kernel stats of the bench run, per-launch PMC means of the force kernel, and
profiles/traffic.json (HBM bytes per launch, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note)."""
import collections, csv, glob, json, shutil, sys

tag, trace_dir, pmc_prefix = sys.argv[1], sys.argv[2], sys.argv[3]
stats = glob.glob(f"{trace_dir}/*/*_kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{tag}_bench_kernel_stats.csv")
summ, dur = {}, []
for f in glob.glob(f"{pmc_prefix}*/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "accel_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    for k, v in agg.items():
        summ[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
kt = sum(dur) / len(dur)
summ["_kernel_seconds_in_pmc_pass"] = kt
summ["_effective_clock_GHz_in_pmc_pass"] = summ["GRBM_GUI_ACTIVE"]["mean"] / 8 / kt / 1e9
fetch = summ["FETCH_SIZE"]["mean"] * 1024 * 2
write = summ["WRITE_SIZE"]["mean"] * 1024
summ["_hbm_bytes_per_launch"] = {"read_FETCH_SIZE_x2": fetch, "write_WRITE_SIZE": write, "total": fetch + write}
summ["_valu_busy_frac"] = summ["SQ_ACTIVE_INST_VALU"]["mean"] * 4 / 1024 / (summ["GRBM_GUI_ACTIVE"]["mean"] / 8)
json.dump(summ, open(f"profiles/{tag}_pmc_accel_kernel.json", "w"), indent=1)
json.dump({"accel_kernel_hbm_bytes_per_launch": fetch + write,
           "source": f"profiles/{tag}_pmc_accel_kernel.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; "
                     "FETCH_SIZE x2 per MI355X_MICROARCH.md HBM section)", "workload": "N=65536, 1 GPU"},
          open("profiles/traffic.json", "w"), indent=1)
# the stats file averages every launch, warm-up (ramping clocks) included; bench.py times the last K
trace = glob.glob(f"{trace_dir}/*/*_kernel_trace.csv")[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(trace))
     if "accel_kernel" in r["Kernel_Name"]]
K = int(sys.argv[4]) if len(sys.argv) > 4 else 100
json.dump({"kernel": "accel_kernel<false>", "launches_total": len(d), "mean_us_all_launches": sum(d) / len(d),
           "timed_launches": K, "mean_us_timed_launches": sum(d[-K:]) / K, "min_us": min(d), "max_us": max(d),
           "note": "timed launches = the last K (bench.py --steps K); earlier ones are warm-up at ramping clocks"},
          open(f"profiles/{tag}_bench_kernel_trace_summary.json", "w"), indent=1)
for r in list(csv.DictReader(open(stats)))[:4]:
    print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, "us", r["Percentage"])
print(json.dumps({k: v for k, v in summ.items() if k.startswith("_")}, indent=1))
print("VALU insts/launch", summ["SQ_INSTS_VALU"]["mean"], "LDS insts", summ["SQ_INSTS_LDS"]["mean"])
