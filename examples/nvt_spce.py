#!/usr/bin/env python3
"""NVT Monte Carlo of SPC/E water the way the reference's `Loop()` runs it (Ewald/main.jl:460-696),
for R independent chains on one GPU: blocks of sweeps with `Adjust!` after every sweep, the
reference's block line (main.jl:662-676), and the O-O radial distribution function at the end
(the intent of gr.jl).

    python3 examples/nvt_spce.py [--replicas 64] [--blocks 5] [--sweeps 20]

Needs an MI355X (no CPU fallback).  Starts from the NIST SPC/E sample configuration 4
(= Ewald/coord750.txt), physical centres of mass.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from metropolismontecarlo_amd import io as mio, observables, structs  # noqa: E402
from metropolismontecarlo_amd.device import Batch, block_line  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=5)
    ap.add_argument("--sweeps", type=int, default=20, help="sweeps (N_mol trial moves) per block")
    ap.add_argument("--temperature", type=float, default=298.15)
    args = ap.parse_args()

    a = mio.load_nist_fixture(4, "unwrapped")
    n_mol, box, r_cut = a["com"].shape[0], a["box"], 10.0
    b = Batch(args.replicas, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], box,
              5.6 / box, structs.factor, r_cut, r_cut)
    b.set_option("device_moves", 1)
    tot = b.potential_ewald()                                   # main.jl:408
    chains = b.new_chains([t["energy"] for t in tot], [t["virial"] for t in tot],
                          dr_max=0.316555789, dphi_max=0.05)    # main.jl:118,73
    for blk in range(1, args.blocks + 1):
        b.run_chains(chains, args.sweeps * n_mol, args.temperature, seed=11234 + 1000 * blk,
                     adjust=True, n_threads=2)
        print(block_line(chains[0], blk, n_mol, box))            # main.jl:667-679, chain 0
    mean = chains["energy"].mean() / n_mol
    err = chains["energy"].std() / n_mol / np.sqrt(args.replicas)
    drift = max(abs(chains["energy"][r] - t["energy"]) / abs(t["energy"])
                for r, t in enumerate(b.potential_ewald()))
    print(f"all {args.replicas} chains: <E>/N = {mean:.2f} +- {err:.2f} K, "
          f"running total vs recompute: {drift:.1e}")
    r, g = observables.normalize_rdf(b.rdf(0, 150), n_mol, box, args.replicas)
    k = int(np.argmax(g))
    print(f"O-O g(r): first peak {g[k]:.2f} at {r[k]:.2f} A; g(4.5 A) = {g[np.searchsorted(r, 4.5)]:.2f}")
    b.close()


if __name__ == "__main__":
    main()
