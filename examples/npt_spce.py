#!/usr/bin/env python3
"""NPT Monte Carlo of SPC/E water: sweeps of trial moves (`Loop()`, Ewald/main.jl:487-644)
interleaved with the volume move the reference states in Ewald/volumeChange.jl:59-147, on ONE
device state -- the trial moves on the batch's move server, the volume move with a device-side
snapshot (a rejection restores coordinates, k-vectors, tables and S(k) bit for bit).

    python3 examples/npt_spce.py [--molecules 750|10000] [--blocks 5] [--sweeps 4] [--pressure-bar 1.0]

Needs an MI355X (no CPU fallback).  750 molecules: the NIST SPC/E sample configuration 4
(= Ewald/coord750.txt; r_cut 10 A needs L >= 20 A); 10000: the reference's cubic start lattice
(BASELINE configs[3]).  LJ tail corrections (energy.jl:514-614) are added to the printed energy and
pressure only, as the reference's driver leaves them out of the acceptance rule
(main.jl:384-385).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from metropolismontecarlo_amd import io as mio, npt, structs  # noqa: E402
from metropolismontecarlo_amd.device import Batch  # noqa: E402

BAR_IN_K_PER_A3 = 1e5 / 1.380649e-23 * 1e-30      # 1 bar = 7.2430e-3 K / A^3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=750, choices=(750, 10000))
    ap.add_argument("--blocks", type=int, default=5)
    ap.add_argument("--sweeps", type=int, default=4, help="sweeps (N_mol trial moves + 1 volume move) per block")
    ap.add_argument("--temperature", type=float, default=298.15)
    ap.add_argument("--pressure-bar", type=float, default=1.0)
    ap.add_argument("--vmax-frac", type=float, default=0.004, help="dV uniform in +- this fraction of V / 2")
    args = ap.parse_args()

    a4 = mio.load_nist_fixture(4, "unwrapped")
    if args.molecules == 750:
        a = a4
    else:
        box, com, coords = mio.cubic_lattice_water(args.molecules, 0.033101144, "spce", seed=11234)
        a = dict(com=com, coords=coords, atype=np.tile([1, 2, 2], args.molecules),
                 charge=np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], args.molecules),
                 eps=a4["eps"], sig=a4["sig"], box=box)
    n_mol, box, r_cut = a["com"].shape[0], float(a["box"]), 10.0
    b = Batch(1, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], box, 5.6 / box,
              structs.factor, r_cut, r_cut)
    b.set_option("device_moves", 1)
    energy = float(b.potential_ewald(as_array=True)["energy"][0])
    P = args.pressure_bar * BAR_IN_K_PER_A3
    vmax = args.vmax_frac * box ** 3
    n_type = np.array([n_mol, 2 * n_mol])                 # atoms of each type (O, H)
    print(f"{n_mol} SPC/E molecules, L = {box:.4f} A, P = {args.pressure_bar} bar, T = {args.temperature} K")
    for blk in range(1, args.blocks + 1):
        energy, st, ns = b.run_npt(args.sweeps, args.temperature, P, vmax, 0.316555789, 0.05,
                                   seed=11234 + blk, energy=energy)
        L = ns["box"]
        rho = n_mol / L ** 3
        e_tail = npt.ener_corr(a["eps"], a["sig"], r_cut, L, n_type)
        acc = (st["trans_accept"] + st["rot_accept"]) / max(st["moves"], 1)
        print(f"Block: {blk:4d}, L: {L:8.4f} A, density: {rho * 18.01528 / 0.602214076:6.4f} g/cm3, "
              f"energy/N: {(energy + e_tail) / n_mol:10.2f} K, move ratio: {acc:4.2f}, "
              f"volume ratio: {ns['vol_accept'] / max(ns['vol_attempt'], 1):4.2f}, "
              f"us/move: {1e3 * st['wall_ms'] / max(st['moves'], 1):6.2f}, ms/volume move: "
              f"{ns['volume_ms'] / max(ns['vol_attempt'], 1):5.2f}")
    check = float(b.potential_ewald(as_array=True)["energy"][0])
    print(f"running total {energy:.6f} K, recomputed {check:.6f} K (Poly/main.jl:232-235)")
    b.close()


if __name__ == "__main__":
    main()
