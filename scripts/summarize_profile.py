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
         "bench_events_avg_us": (bench.get("one_stream") or {}).get("avg_launch_us") or bench["roofline"]["avg_launch_us"],
         "note": "trace and PMC passes run `bench.py --streams 1` (every launch alone on the GPU: its duration is "
                 "its cost); the default's two streams overlap launches -- their spans are in *_two_stream_kernel_stats.csv"}
try:
    shutil.copy(newest("trace2/*/*kernel_stats.csv"), os.path.join(DST, f"{TAG}_two_stream_kernel_stats.csv"))
except SystemExit:
    pass

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
# the roofline object's companions: what binds the kernel beside bytes
extras = {"kernel": KERNEL, "moves_per_launch": moves_per_launch}
if "SQ_INSTS_VALU" in pmc:
    extras["valu_insts_per_move"] = pmc["SQ_INSTS_VALU"]["per_dispatch"] / moves_per_launch
if wc and "SQ_ACTIVE_INST_VALU" in pmc and w:
    n_simd = 256 * 4
    waves_per_simd = min(w / n_simd, 8.0)
    extras["waves_per_simd"] = waves_per_simd
    # SQ_WAVE_CYCLES sums the resident waves' cycles; the vector pipe of a SIMD serves all of its
    # waves, so its busy share is issue cycles x waves per SIMD / wave cycles
    extras["valu_busy_frac"] = pmc["SQ_ACTIVE_INST_VALU"]["per_dispatch"] * waves_per_simd / wc
    # (SQ_WAVE_CYCLES counts quad-cycles summed over the resident waves: x 4 / waves / duration is
    # the shader clock the counters imply while the kernel runs)
    extras["clock_ghz"] = wc * 4.0 / (n_simd * waves_per_simd) / (trace["avg_us"] * 1e3)
if "SQ_INSTS_SALU" in pmc:
    extras["salu_insts_per_move"] = pmc["SQ_INSTS_SALU"]["per_dispatch"] / moves_per_launch
if wc and "SQ_WAIT_ANY" in pmc:
    extras["wait_frac"] = pmc["SQ_WAIT_ANY"]["per_dispatch"] / wc
if "SQ_LDS_IDX_ACTIVE" in pmc and "clock_ghz" in extras:
    # LDS-array cycles summed over the 256 compute units against the cycles of the launch
    cu_cycles = 256 * extras["clock_ghz"] * 1e3 * trace["avg_us"]
    extras["lds_busy_frac"] = pmc["SQ_LDS_IDX_ACTIVE"]["per_dispatch"] / cu_cycles
    if "SQ_LDS_BANK_CONFLICT" in pmc:
        extras["lds_conflict_frac"] = pmc["SQ_LDS_BANK_CONFLICT"]["per_dispatch"] / pmc["SQ_LDS_IDX_ACTIVE"]["per_dispatch"]
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    hbm = (2 * pmc["FETCH_SIZE"]["per_dispatch"] + pmc["WRITE_SIZE"]["per_dispatch"]) * 1024
    extras["hbm_traffic_gbs"] = hbm / (trace["avg_us"] * 1e-6) / 1e9
    extras["hbm_traffic_frac"] = extras["hbm_traffic_gbs"] / 8000.0
json.dump({"trace": trace, "pmc": pmc, "derived": derived, "roofline_extras": extras},
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
    # ... and `binding` from this session's counters (the run quoted the previous profile's)
    rf = bench["roofline"]
    t = rf["avg_launch_us"] * 1e-6
    mpl = rf["moves_per_launch"]
    src = (f"profiles/{TAG}_default_pmc_summary.json (rocprofv3 --pmc passes of the same profiling session; "
           "filled in by scripts/summarize_profile.py, not measured by this run)")
    pat = os.path.join(ROOT, "gpurun_out", "gather_bw.json")
    if os.path.exists(pat):
        shutil.copy(pat, os.path.join(DST, f"{TAG}_access_pattern_bw.json"))
    # recompute with bench.py's own code, which now finds this session's files under profiles/
    sys.path.insert(0, ROOT)
    import argparse
    import bench as bench_mod
    args = argparse.Namespace(kernel=3, no_events=False)
    def again(r, streams, elapsed_us_per_launch):
        st = {"launches": r["launches"], "moves": r["moves_per_launch"] * r["launches"],
              "timed_launches": r["launches_timed_with_events"],
              "kernel_ms": (r.get("launch_span_us") or r["avg_launch_us"]) * 1e-3 * r["launches_timed_with_events"]}
        res = {"st": st, "elapsed": elapsed_us_per_launch * 1e-6 * r["launches"], "streams": streams}
        return bench_mod.roofline_object(res, bench["config"]["replicas_per_gpu"], args,
                                         {"groups": bench["config"]["groups_per_gpu"]}, 750, 30.0, 1)
    new = again(rf, bench["config"].get("streams_per_gpu", 1), rf["avg_launch_us"])
    if new and "binding" in new:
        rf["binding"] = new["binding"]
        rf["binding"]["counters_source"] = src
        one = bench.get("one_stream")
        if one and one.get("avg_launch_us"):
            b1 = dict(rf)
            b1.update(avg_launch_us=one["avg_launch_us"], launch_span_us=one["avg_launch_us"])
            n1 = again(b1, 1, one["avg_launch_us"])
            if n1 and "binding" in n1:
                one["binding_frac"], one["binding_bound"] = n1["binding"]["frac"], n1["binding"]["bound"]
    json.dump(bench, open(os.path.join(DST, f"{TAG}_default_bench.json"), "w"))
# ---- the persistent move server of one chain (BASELINE configs[1]): counters of its longest dispatch
def server_summary():
    steps = 20000  # scripts/profile_round.sh
    out = {"command": "bench.py --no-cpu --no-secondary --replicas 1 --steps 20000 --warmup 300",
           "steps_of_the_dispatch": steps}
    try:
        tr = list(csv.DictReader(open(newest("trace_server/*/*kernel_trace.csv"))))
    except SystemExit:
        return None
    srv = [r for r in tr if "k_move_server" in r["Kernel_Name"]]
    if not srv:
        return None
    big = max(srv, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    dur = int(big["End_Timestamp"]) - int(big["Start_Timestamp"])
    out["kernel"] = big["Kernel_Name"].split("(")[0]
    out["workgroups"] = int(big["Grid_Size_X"]) // int(big["Workgroup_Size_X"])
    out["waves_per_workgroup"] = int(big["Workgroup_Size_X"]) // 64
    out["us_per_step_in_kernel"] = dur / 1e3 / steps
    cnt = {}
    for name in ("pmc_server1", "pmc_server2"):
        rows = [r for r in csv.DictReader(open(newest(f"{name}/*/*counter_collection.csv")))
                if "k_move_server" in r["Kernel_Name"]]
        by_disp = collections.defaultdict(dict)
        for r in rows:
            by_disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        key = "SQ_WAVE_CYCLES" if name == "pmc_server1" else "SQ_WAIT_ANY"
        best = max(by_disp.values(), key=lambda d: d.get(key, 0.0))
        cnt.update(best)
    out["counters_of_the_dispatch"] = cnt
    if "SQ_INSTS_VALU" in cnt:
        out["valu_insts_per_step"] = cnt["SQ_INSTS_VALU"] / steps
        out["salu_insts_per_step"] = cnt.get("SQ_INSTS_SALU", 0.0) / steps
        out["lds_insts_per_step"] = cnt.get("SQ_INSTS_LDS", 0.0) / steps
    if "SQ_WAVE_CYCLES" in cnt and "SQ_ACTIVE_INST_VALU" in cnt:
        out["valu_issue_share_of_wave_cycles"] = cnt["SQ_ACTIVE_INST_VALU"] / cnt["SQ_WAVE_CYCLES"]
        out["wait_share_of_wave_cycles"] = cnt.get("SQ_WAIT_ANY", 0.0) / cnt["SQ_WAVE_CYCLES"]
    return out


srv = server_summary()
if srv:
    json.dump(srv, open(os.path.join(DST, f"{TAG}_server_pmc_summary.json"), "w"), indent=1)
    print({k: v for k, v in srv.items() if k != "counters_of_the_dispatch"})
print(json.dumps(trace))
print({k: round(v["per_dispatch"], 1) for k, v in pmc.items()})
print({k: traffic[k] for k in ("bytes_per_launch", "bytes_per_move")})
