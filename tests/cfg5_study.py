"""BASELINE.json configs[4] / SURVEY.md 8(d) "cfg5": TIP3P-parameter water, 5000 molecules on the
reference's cubic start lattice, Wolf summation (energy.jl:864-943) vs Ewald (energy.jl:946-1032),
fp32 vs fp64 -- measured against the ORACLE's fp64 (test infrastructure: this module lives under
tests/ because it imports oracle/).

fp64: the product path (mmc_potential_ewald / mmc_potential_wolf / mmc_trial_move); fp32 and
"mixed" (fp32 arithmetic, fp64 accumulators): the study kernels (csrc/mmc_study.hpp).  Every
scripted move starts from the same configuration (translation (-1/2,1/2)^3 x
0.316555789 A, rotation +-0.05 rad about a uniform axis).

A tolerance STUDY of single precision (it will not hold 1e-6 on dU), and a parity GATE for the
fp64 path at this configuration's size.  Used by tests/test_gpu_at_size.py and by
scripts/precision_study.py, which writes profiles/roundN_cfg5_precision_study.json.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MOL = 5000
RHO = 5000 / 53.257 ** 3        # 18^3 lattice, first 5000 sites, L = 53.257 A
Q_O, Q_H = -0.834, 0.417        # topol.top / water.top:25-27
SIG_OO, EPS_OO = 3.15061, 0.6364 / 0.0083144621
RCUT, SEED = 10.0, 11234


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def _err(x, ref):
    scale = np.maximum(np.abs(ref), 1.0)
    return {"max_abs_err_K": float(np.abs(x - ref).max()),
            "rms_abs_err_K": float(np.sqrt(np.mean((x - ref) ** 2))),
            "max_rel_err(|dU|>=1K floor)": float((np.abs(x - ref) / scale).max())}


def run(n_moves=10000, n_mol=N_MOL):
    from metropolismontecarlo_amd import io as mio, moves, structs
    from metropolismontecarlo_amd.device import Context
    from oracle import oracle as orc

    box, com, coords = mio.cubic_lattice_water(n_mol, RHO, "tip3p", seed=SEED)
    first = 3 * np.arange(n_mol, dtype=np.int64) + 1
    eps = np.array([[EPS_OO, 0.0], [0.0, 0.0]])
    sig = np.array([[SIG_OO, 0.0], [0.0, 0.0]])
    atype = np.tile([1, 2, 2], n_mol)
    charge = np.tile([Q_O, Q_H, Q_H], n_mol)
    s = orc.System(com, first, first + 2, coords, atype, charge, eps, sig, box)
    ew = orc.Ewald(5.6 / box, 5, 27, box)
    t0 = time.perf_counter()
    eo = orc.potential_ewald(s, ew, RCUT, RCUT)          # also fills ew.sumQExpOld / New
    wo = orc.potential_wolf(s, ew, RCUT, RCUT, literal_prefactor=False)
    oracle_totals_s = time.perf_counter() - t0
    ref = {"lj": eo["lj"], "real": eo["real"], "recip": eo["recip"], "ewald_self": eo["self"],
           "wolf_const": wo["self"], "ewald_total": eo["energy"], "wolf_total": wo["energy"]}

    ctx = Context(0)
    try:
        ctx.upload_system(com, first, first + 2, coords, atype, charge, eps, sig, box)
        ctx.prepare_ewald(5.6 / box, 5, 27, box, structs.factor)
        e64 = ctx.potential_ewald(RCUT, RCUT)
        w64 = ctx.potential_wolf(RCUT, RCUT)
        p64 = {"lj": e64["lj"], "real": e64["real"], "recip": e64["recip"],
               "ewald_self": e64["self"], "wolf_const": w64["self"],
               "ewald_total": e64["energy"], "wolf_total": w64["energy"]}
        totals = {"oracle_fp64": ref, "fp64": p64,
                  "fp64_vs_oracle_rel": {k: _rel(p64[k], ref[k]) for k in ref}}
        for name, mixed in (("fp32", False), ("mixed", True)):
            t = ctx.study_f32_total(RCUT, RCUT, mixed)
            ewt = t["lj"] + t["real"] + t["recip"] + ref["ewald_self"]
            wot = t["lj"] + t["real"] + ref["wolf_const"]
            totals[name] = {"lj": t["lj"], "real": t["real"], "recip": t["recip"],
                            "ewald_total": ewt, "wolf_total": wot, "n_overlap": t["n_overlap"],
                            "rel_err_vs_oracle": {"lj": _rel(t["lj"], ref["lj"]),
                                                  "real": _rel(t["real"], ref["real"]),
                                                  "recip": _rel(t["recip"], ref["recip"]),
                                                  "ewald_total": _rel(ewt, ref["ewald_total"]),
                                                  "wolf_total": _rel(wot, ref["wolf_total"])}}
        totals["wolf_vs_ewald_fp64"] = {"abs": ref["wolf_total"] - ref["ewald_total"],
                                        "rel": _rel(ref["wolf_total"], ref["ewald_total"]),
                                        "per_molecule_K": (ref["wolf_total"] - ref["ewald_total"]) / n_mol}

        rng = np.random.default_rng(SEED)
        t0 = time.perf_counter()
        rows = []
        for _ in range(n_moves):
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
            do, ovo = orc.trial_move(i, s, ew, RCUT, RCUT, c_new, a_new)
            ew.sumQExpNew = ew.sumQExpOld.copy()                       # rejected (main.jl:628)
            d64, ov = ctx.trial_move(i, c_new, a_new, RCUT, RCUT)
            ctx.reject_move()
            d32, ov32 = ctx.study_f32_move(i, c_new, a_new, RCUT, RCUT, False)
            dmx, ovmx = ctx.study_f32_move(i, c_new, a_new, RCUT, RCUT, True)
            rows.append((do[0], do[1], do[2], d64[0], d64[1], d64[2], d32[0], d32[1], d32[2],
                         dmx[0], dmx[1], dmx[2], ovo, ov, ov32, ovmx))
        elapsed = time.perf_counter() - t0
    finally:
        ctx.close()
    r = np.array(rows, dtype=float)
    ew_o, wo_o = r[:, 0] + r[:, 1] + r[:, 2], r[:, 0] + r[:, 1]
    du = {}
    for name, col, ovc in (("fp64", 3, 13), ("fp32", 6, 14), ("mixed", 9, 15)):
        ewd, wod = r[:, col] + r[:, col + 1] + r[:, col + 2], r[:, col] + r[:, col + 1]
        du[name] = {"ewald": _err(ewd, ew_o), "wolf": _err(wod, wo_o),
                    "per_term_max_abs_err_K": {"lj": float(np.abs(r[:, col] - r[:, 0]).max()),
                                               "real": float(np.abs(r[:, col + 1] - r[:, 1]).max()),
                                               "recip": float(np.abs(r[:, col + 2] - r[:, 2]).max())},
                    "overlap_flags_differ": int((r[:, 12] != r[:, ovc]).sum())}
    du["oracle_dU_scale"] = {"rms_ewald_K": float(np.sqrt(np.mean(ew_o ** 2))),
                             "rms_recip_K": float(np.sqrt(np.mean(r[:, 2] ** 2))),
                             "wolf_minus_ewald_rms_K": float(np.sqrt(np.mean(r[:, 2] ** 2)))}
    return {"config": {"n_mol": n_mol, "box": box, "kappa": 5.6 / box, "r_cut": RCUT,
                       "model": "TIP3P charges/LJ (topol.top), geometry of tip3p.pdb:3-5",
                       "moves": n_moves, "seconds": elapsed, "oracle_totals_seconds": oracle_totals_s,
                       "reference": "oracle/mmc_oracle.c (fp64 CPU restatement)"},
            "totals": totals, "dU": du, "kT_at_298K": 298.15,
            "note": "Wolf dU = dLJ + dReal (no reciprocal term, main.jl:580-590), Ewald dU = dLJ + "
                    "dReal + dRecip; every error is against the oracle's fp64 value; acceptance "
                    "depends on dU / 298.15 K"}
