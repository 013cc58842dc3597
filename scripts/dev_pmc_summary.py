#!/usr/bin/env python3
"""Per-dispatch averages of the dominant move kernel from a scripts/dev_pmc.sh output directory."""
import collections, csv, glob, json, os, sys
src = sys.argv[1]
def newest(p):
    f = sorted(glob.glob(os.path.join(src, p)), key=os.path.getmtime)
    return f[-1] if f else None
out = {}
tr = newest("trace/*/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    if "k_move_eval" in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
main = max(dur, key=lambda g: sum(dur[g]))
out["kernel"], out["grid_threads"] = main
out["calls"] = len(dur[main]); out["avg_us"] = sum(dur[main]) / len(dur[main]) / 1e3
for name in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    f = newest(f"{name}/*/*counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if main[0] in r["Kernel_Name"]:
            agg[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg: continue
    g = max(agg, key=lambda k: sum(len(v) for v in agg[k].values()))
    for c, v in agg[g].items():
        out[c] = sum(v) / len(v)
w = out.get("SQ_WAVES", 0)
if w:
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
        if c in out: out[c + "_per_wave"] = out[c] / w
wc = out.get("SQ_WAVE_CYCLES")
if wc:
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if c in out: out[c + "_share"] = out[c] / wc
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
print(json.dumps(out, indent=1))
