#!/usr/bin/env python3
"""BASELINE.json configs[4] / SURVEY.md 8(d) "cfg5": TIP3P-parameter water, 5000 molecules on the
reference's cubic start lattice, Wolf summation vs Ewald, fp32 vs fp64.

A tolerance STUDY, not a gate ("the fp32 sweep of config 5 will not hold 1e-6
on dU").  fp64 numbers come from the product path (mmc_potential_ewald / mmc_potential_wolf /
mmc_trial_move); fp32 and mixed (fp32 arithmetic, fp64 accumulators) numbers from the study
kernels (csrc/mmc_study.hpp).  Every scripted move starts from the same configuration.

    python3 scripts/precision_study.py [--moves 10000] [--out gpurun_out/precision_study.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from metropolismontecarlo_amd import io as mio, moves, structs  # noqa: E402
from metropolismontecarlo_amd.device import Context  # noqa: E402

N_MOL = 5000
RHO = 5000 / 53.257 ** 3        # 18^3 lattice, first 5000 sites, L = 53.257 A
Q_O, Q_H = -0.834, 0.417        # topol.top / SURVEY 8(d)
SIG_OO, EPS_OO = 3.15061, 0.6364 / 0.0083144621
RCUT, SEED = 10.0, 11234


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--moves", type=int, default=10000)
    ap.add_argument("--n-mol", type=int, default=N_MOL)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "precision_study.json"))
    args = ap.parse_args()

    n_mol = args.n_mol
    box, com, coords = mio.cubic_lattice_water(n_mol, RHO, "tip3p", seed=SEED)
    first = 3 * np.arange(n_mol, dtype=np.int64) + 1
    eps = np.array([[EPS_OO, 0.0], [0.0, 0.0]])
    sig = np.array([[SIG_OO, 0.0], [0.0, 0.0]])
    charge = np.tile([Q_O, Q_H, Q_H], n_mol)
    ctx = Context(0)
    ctx.upload_system(com, first, first + 2, coords, np.tile([1, 2, 2], n_mol), charge, eps, sig, box)
    ctx.prepare_ewald(5.6 / box, 5, 27, box, structs.factor)

    # ---- totals: a11 (Ewald) and a12 (Wolf) ----
    e64 = ctx.potential_ewald(RCUT, RCUT)
    w64 = ctx.potential_wolf(RCUT, RCUT)
    totals = {"fp64": {"lj": e64["lj"], "real": e64["real"], "recip": e64["recip"],
                       "ewald_self": e64["self"], "wolf_const": w64["self"],
                       "ewald_total": e64["energy"], "wolf_total": w64["energy"]}}
    for name, mixed in (("fp32", False), ("mixed", True)):
        t = ctx.study_f32_total(RCUT, RCUT, mixed)
        ew = t["lj"] + t["real"] + t["recip"] + e64["self"]
        wo = t["lj"] + t["real"] + w64["self"]
        totals[name] = {"lj": t["lj"], "real": t["real"], "recip": t["recip"],
                        "ewald_total": ew, "wolf_total": wo,
                        "rel_err": {"lj": rel(t["lj"], e64["lj"]), "real": rel(t["real"], e64["real"]),
                                    "recip": rel(t["recip"], e64["recip"]),
                                    "ewald_total": rel(ew, e64["energy"]),
                                    "wolf_total": rel(wo, w64["energy"])}}
    totals["wolf_vs_ewald_fp64"] = {"abs": w64["energy"] - e64["energy"],
                                    "rel": rel(w64["energy"], e64["energy"]),
                                    "per_molecule_K": (w64["energy"] - e64["energy"]) / n_mol}

    # ---- per-move dU over scripted moves (translation (-1/2,1/2)^3 * 0.316555789 A,
    # rotation +-0.05 rad about a uniform axis), all from the same configuration ----
    rng = np.random.default_rng(SEED)
    t0 = time.perf_counter()
    rows = []
    for _ in range(args.moves):
        i = int(rng.integers(1, n_mol + 1))
        c_old, a_old = com[i - 1], coords[3 * (i - 1):3 * i]
        if rng.random() < 0.5:
            c_new = moves.random_translate_vector(0.316555789, c_old, box, rng)
            a_new = a_old + (c_new - c_old)
        else:
            axis = moves.random_vector(rng)
            ang = (2.0 * rng.random() - 1.0) * 0.05
            c, s_, t = np.cos(ang), np.sin(ang), 1 - np.cos(ang)
            x, y, z = axis
            Rm = np.array([[t * x * x + c, t * x * y - s_ * z, t * x * z + s_ * y],
                           [t * x * y + s_ * z, t * y * y + c, t * y * z - s_ * x],
                           [t * x * z - s_ * y, t * y * z + s_ * x, t * z * z + c]])
            c_new = c_old.copy()
            a_new = c_old + (a_old - c_old) @ Rm.T
        d64, ov = ctx.trial_move(i, c_new, a_new, RCUT, RCUT)
        ctx.reject_move()
        d32, ov32 = ctx.study_f32_move(i, c_new, a_new, RCUT, RCUT, False)
        dmx, ovmx = ctx.study_f32_move(i, c_new, a_new, RCUT, RCUT, True)
        rows.append((d64[0], d64[1], d64[2], d32[0], d32[1], d32[2], dmx[0], dmx[1], dmx[2],
                     ov, ov32, ovmx))
    elapsed = time.perf_counter() - t0
    r = np.array(rows, dtype=float)
    du = {}
    for name, col in (("fp32", 3), ("mixed", 6)):
        ew64, wo64 = r[:, 0] + r[:, 1] + r[:, 2], r[:, 0] + r[:, 1]
        ew, wo = r[:, col] + r[:, col + 1] + r[:, col + 2], r[:, col] + r[:, col + 1]
        scale_e, scale_w = np.maximum(np.abs(ew64), 1.0), np.maximum(np.abs(wo64), 1.0)
        du[name] = {
            "ewald": {"max_abs_err_K": float(np.abs(ew - ew64).max()),
                      "rms_abs_err_K": float(np.sqrt(np.mean((ew - ew64) ** 2))),
                      "max_rel_err(|dU|>=1K floor)": float((np.abs(ew - ew64) / scale_e).max())},
            "wolf": {"max_abs_err_K": float(np.abs(wo - wo64).max()),
                     "rms_abs_err_K": float(np.sqrt(np.mean((wo - wo64) ** 2))),
                     "max_rel_err(|dU|>=1K floor)": float((np.abs(wo - wo64) / scale_w).max())},
            "per_term_max_abs_err_K": {"lj": float(np.abs(r[:, col] - r[:, 0]).max()),
                                       "real": float(np.abs(r[:, col + 1] - r[:, 1]).max()),
                                       "recip": float(np.abs(r[:, col + 2] - r[:, 2]).max())},
            "overlap_flags_differ": int((r[:, 9] != r[:, 10 if col == 3 else 11]).sum()),
        }
    du["fp64_dU_scale"] = {"rms_ewald_K": float(np.sqrt(np.mean((r[:, 0] + r[:, 1] + r[:, 2]) ** 2))),
                           "rms_recip_K": float(np.sqrt(np.mean(r[:, 2] ** 2))),
                           "wolf_minus_ewald_rms_K": float(np.sqrt(np.mean(r[:, 2] ** 2)))}
    out = {"config": {"n_mol": n_mol, "box": box, "kappa": 5.6 / box, "r_cut": RCUT,
                      "model": "TIP3P charges/LJ (topol.top), geometry of tip3p.pdb:3-5",
                      "moves": args.moves, "seconds": elapsed},
           "totals": totals, "dU": du,
           "kT_at_298K": 298.15,
           "note": "Wolf dU = dLJ + dReal (no reciprocal term), Ewald dU = dLJ + dReal + dRecip; "
                   "acceptance depends on dU / 298.15 K"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
