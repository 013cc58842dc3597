#!/usr/bin/env python3
"""Per-dispatch averages of every kernel of scripts/m2_profile.sh's passes (largest grid of each)."""
import collections, csv, glob, json, os, sys
src = sys.argv[1]
def newest(p):
    f = sorted(glob.glob(os.path.join(src, p)), key=os.path.getmtime)
    return f[-1] if f else None
dur = collections.defaultdict(list)
for r in csv.DictReader(open(newest("trace/*/*kernel_trace.csv"))):
    key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for (k, g), v in dur.items():
    out.setdefault(k, {})[g] = {"calls": len(v), "avg_us": sum(v) / len(v) / 1e3}
pmc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for name in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    f = newest(f"{name}/*/*counter_collection.csv")
    if not f: continue
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"].split("(")[0]][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in out:
    for g in out[k]:
        c = {n: sum(v) / len(v) for n, v in pmc.get(k, {}).get(g, {}).items()}
        w = c.get("SQ_WAVES")
        if w:
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
                if n in c: c[n + "_per_wave"] = c[n] / w
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if n in c: c[n + "_share"] = c[n] / wc
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["hbm_bytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        out[k][g].update(c)
keep = {k: v for k, v in out.items() if any(x in k for x in ("k_total", "k_recip", "k_atom", "k_charge", "k_mol", "k_rescale", "k_kvec", "k_build"))}
print(json.dumps(keep, indent=1))
