#!/usr/bin/env python3
"""Turn gpurun_out/round_profile/ (written by scripts/profile_round.sh) into the files under
profiles/: the bench line, the rocprofv3 kernel stats, the per-dispatch PMC averages of the
dominant kernel and the HBM traffic file bench.py reads.

Only dispatches of the full-size launch are averaged (the grid of the main workload), so that
launches of the small secondary configurations do not dilute the per-launch figures.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round_profile")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "round1"
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "k_move_eval_wave"


def newest(pattern):
    files = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"nothing matches {pattern}")
    return files[-1]


bench = json.load(open(os.path.join(SRC, "bench_default.json")))
moves_per_launch = int(bench["roofline"]["moves_per_launch"])
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, f"{TAG}_default_bench.json"))
shutil.copy(newest("trace/*/*kernel_stats.csv"), os.path.join(DST, f"{TAG}_default_kernel_stats.csv"))

# kernel trace: average duration of the full-size launches only
dur = collections.defaultdict(list)
for r in csv.DictReader(open(newest("trace/*/*kernel_trace.csv"))):
    if KERNEL in r["Kernel_Name"]:
        dur[int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
main_grid = max(dur, key=lambda g: sum(dur[g]))
trace = {"kernel": KERNEL, "grid_threads": main_grid, "calls": len(dur[main_grid]),
         "avg_us": sum(dur[main_grid]) / len(dur[main_grid]) / 1e3,
         "bench_events_avg_us": bench["roofline"]["avg_launch_us"]}

pmc = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(f"{name}/*/*counter_collection.csv"))):
        if KERNEL in r["Kernel_Name"]:
            agg[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    grid = max(agg, key=lambda g: sum(len(v) for v in agg[g].values()))   # the main launch shape
    for c, v in agg[grid].items():
        pmc[c] = {"per_dispatch": sum(v) / len(v), "dispatches": len(v), "grid_threads": grid}
derived = {}
w = pmc.get("SQ_WAVES", {}).get("per_dispatch")
if w:
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
        if c in pmc:
            derived[c + "_per_move"] = pmc[c]["per_dispatch"] / moves_per_launch
wc = pmc.get("SQ_WAVE_CYCLES", {}).get("per_dispatch")
if wc:
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if c in pmc:
            derived[c + "_share_of_wave_cycles"] = pmc[c]["per_dispatch"] / wc
if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
    h, m = pmc["TCC_HIT_sum"]["per_dispatch"], pmc["TCC_MISS_sum"]["per_dispatch"]
    derived["l2_hit_rate"] = h / (h + m)
json.dump({"trace": trace, "pmc": pmc, "derived": derived},
          open(os.path.join(DST, f"{TAG}_default_pmc_summary.json"), "w"), indent=1)

fetch = pmc["FETCH_SIZE"]["per_dispatch"]
write = pmc["WRITE_SIZE"]["per_dispatch"]
traffic = {
    "kernel": KERNEL,
    "moves_per_launch": moves_per_launch,
    "FETCH_SIZE_KB_per_launch": fetch,
    "WRITE_SIZE_KB_per_launch": write,
    "correction": "MI355X_MICROARCH.md: counters are in KB and FETCH_SIZE under-reports 2x on gfx950 -> "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
    "bytes_per_launch": (2 * fetch + write) * 1024,
    "bytes_per_move": (2 * fetch + write) * 1024 / moves_per_launch,
}
json.dump(traffic, open(os.path.join(DST, f"{TAG}_traffic.json"), "w"), indent=1)
# the bench line of this profile ran before the counters were reduced: it echoes the traffic file
# of the previous profile, or none -- give the committed copy the figure of its own session
if bench.get("roofline") and bench["roofline"].get("kernel") == KERNEL:
    bench["roofline"]["traffic"] = traffic["bytes_per_launch"]
    bench["roofline"]["traffic_source"] = (f"profiles/{TAG}_traffic.json: the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                           "passes of the same profiling session (scripts/profile_round.sh); "
                                           "filled in by scripts/summarize_profile.py, not measured by this run")
    json.dump(bench, open(os.path.join(DST, f"{TAG}_default_bench.json"), "w"))
print(json.dumps(trace))
print({k: round(v["per_dispatch"], 1) for k, v in pmc.items()})
print({k: traffic[k] for k in ("bytes_per_launch", "bytes_per_move")})
