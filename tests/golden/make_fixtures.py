#!/usr/bin/env python
"""Generate the parity fixtures under tests/golden/.  Run in the build container only:

    python tests/golden/make_fixtures.py

Inputs (read as DATA, nothing is executed from the reference):
    /root/reference/Ewald/spce_sample_config_periodic{1..4}.txt, /root/reference/Ewald/coord750.txt
    -- the NIST SPC/E sample configurations (public NIST data redistributed by the reference;
    format: line 1 box lengths, line 2 N_mol, then `index x y z element`).

Outputs:
    spce_nist.npz       box_k, xyz_k (n,3), is_oxygen_k for k = 1..4 (coord750.txt is checked to be
                        numerically identical to config 4 and not stored twice)
    golden_oracle.json  values of the CPU oracle (oracle/mmc_oracle.c) on those configurations in
                        both COM conventions (see metropolismontecarlo_amd.io.nist_system):
                        totals, per-molecule LJ_poly_dU / EwaldReal for a few molecules, and a
                        scripted list of trial moves with dU terms and S(k) checksums.
The reference itself cannot be run here (Julia is not installed), so these are ORACLE values: they
pin the oracle against regressions; the external pins are the NIST energies in test_oracle.py.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from metropolismontecarlo_amd import io as mio  # noqa: E402  (host-side loader, no GPU needed)
from oracle import oracle as orc  # noqa: E402

REF = "/root/reference/Ewald"


def scripted_moves(n_mol, box, com, coords, n_moves=12, seed=20261004):
    """Deterministic proposals: alternating translations (|d| <= 0.158 A per axis, the reference's
    dr_max/2) and rigid rotations (<= 0.05 rad) about the centre of mass."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_moves):
        i = int(rng.integers(1, n_mol + 1))
        c = com[i - 1].copy()
        at = coords[3 * (i - 1):3 * i].copy()
        if k % 2 == 0:
            d = (rng.random(3) - 0.5) * 0.316555789
            cn = c + d
            for ax in range(3):  # PBC (boundaries.jl:16-26)
                if cn[ax] > box:
                    cn[ax] -= box
                if cn[ax] < 0:
                    cn[ax] += box
            an = at + (cn - c)
        else:
            axis = rng.normal(size=3)
            axis /= np.linalg.norm(axis)
            ang = (2 * rng.random() - 1) * 0.05
            K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
            Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
            cn = c
            an = c + (at - c) @ Rm.T
        out.append((i, cn, an))
    return out


def main():
    data = {}
    golden = {}
    for k in range(1, 5):
        box, xyz, is_o = mio.read_nist_text(f"{REF}/spce_sample_config_periodic{k}.txt")
        data[f"box_{k}"] = box
        data[f"xyz_{k}"] = xyz
        data[f"is_oxygen_{k}"] = is_o
    b750, x750, o750 = mio.read_nist_text(f"{REF}/coord750.txt")
    assert b750 == data["box_4"] and np.array_equal(x750, data["xyz_4"]) and np.array_equal(o750, data["is_oxygen_4"])
    np.savez_compressed(os.path.join(HERE, "..", "..", "metropolismontecarlo_amd", "data",
                                     "spce_nist.npz"), **data)  # ships with the package

    for k in range(1, 5):
        for variant in ("reference", "unwrapped"):
            a = mio.nist_system(data[f"box_{k}"], data[f"xyz_{k}"], data[f"is_oxygen_{k}"], variant)
            s = orc.System(a["com"], a["first_atom"], a["last_atom"], a["coords"], a["atype"],
                           a["charge"], a["eps"], a["sig"], a["box"])
            ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
            g = {"n_mol": s.n_mol, "box": s.box, "kappa": ew.kappa, "factor": ew.factor,
                 "nkvecs": ew.NKVECS}
            g["totals_ewald"] = orc.potential_ewald(s, ew, 10.0, 10.0)
            g["sum_abs_S"] = float(np.abs(ew.sumQExpNew).sum())
            g["totals_wolf"] = orc.potential_wolf(s, ew, 10.0, 10.0, literal_prefactor=(k == 1))
            mols = sorted({1, 2, s.n_mol // 2, s.n_mol})
            g["per_mol"] = {}
            for i in mols:
                lj = orc.lj_poly_du(i, s, 10.0)
                er = orc.ewald_real(i, s, ew.kappa, 10.0)
                g["per_mol"][str(i)] = {"lj": lj, "real": [er[0], int(er[1])]}
            # scripted chain: accept every second non-overlapping move so commit AND rollback occur
            moves = []
            for n, (i, cn, an) in enumerate(scripted_moves(s.n_mol, s.box, s.com, s.coords)):
                d, ov = orc.trial_move(i, s, ew, 10.0, 10.0, cn, an)
                accept = (n % 3 != 2) and not ov
                if accept:
                    s.com[i - 1] = cn
                    s.coords[3 * (i - 1):3 * i] = an
                    ew.sumQExpOld = ew.sumQExpNew.copy()   # main.jl:621
                else:
                    ew.sumQExpNew = ew.sumQExpOld.copy()   # main.jl:628
                moves.append({"mol": i, "com_new": cn.tolist(), "atoms_new": an.tolist(),
                              "d": d.tolist(), "overlap": int(ov), "accept": int(accept),
                              "sum_abs_S_old": float(np.abs(ew.sumQExpOld).sum())})
            g["moves"] = moves
            golden[f"config{k}_{variant}"] = g
            print(f"config{k} {variant}: E={g['totals_ewald']['energy']:.9e} "
                  f"recip={g['totals_ewald']['recip']:.9e} self={g['totals_ewald']['self']:.9e}")
    with open(os.path.join(HERE, "golden_oracle.json"), "w") as fh:
        json.dump(golden, fh, indent=1)


if __name__ == "__main__":
    main()
