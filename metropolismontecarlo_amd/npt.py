"""NPT volume move around the device kernels -- host logic of SURVEY.md section 8(f) row 2.

The reference has no executable volume move: Ewald/volumeChange.jl is one docstring holding
Fortran-flavoured pseudo-code (`MC_vol`).  This module follows that text line by line:

    vol_new = vol_old + (rand() - 0.5) * vmax                    volumeChange.jl:59
    L_new   = vol_new^(1/3);  f = L_new / L_old                  :60-62
    COMs scale by f, atoms translate rigidly with their molecule :64-80   -> mmc_volume_change
    total energy over all molecule pairs i < j at L_new          :91-111  -> mmc_potential_ewald
    test = exp(-beta (P dV - N ln(V_new/V_old)/beta + dE))       :129-130
    accept if rand() < test                                      :132

For Ewald electrostatics "the total energy at L_new" implies what BASELINE.json config 4 calls the
full recompute of k-vectors: kappa = alpha / L_new (Ewald/main.jl:290-291), new kxyz/cfac
(PrepareEwaldVariables), new S(k) (RecipLong) -- all inside the two device calls.  Energies are in
K, so beta = 1/T and the pressure is in K / A^3.
"""
import math

import numpy as np


def VolumeChange(ctx, energy_old, box, n_mol, pressure, temperature, vmax, lj_rcut, qq_rcut, rng,
                 alpha=5.6):
    """One volume move of the system held by `ctx` (device.Context).

    Returns (accepted, box, energy, totals).  On rejection the previous coordinates, tables and
    S(k) are restored exactly -- from a copy the DEVICE took before the move (mmc_volume_trial /
    mmc_volume_reject: one launch each way, nothing crosses PCIe) -- as `MC_vol` only overwrites
    its state when the move is accepted (volumeChange.jl:132-147); acceptance costs nothing."""
    vol_old = box ** 3
    vol_new = vol_old + (rng.random() - 0.5) * vmax
    if vol_new <= 0.0:
        return False, box, energy_old, None
    L_new = vol_new ** (1.0 / 3.0)
    if lj_rcut > L_new / 2 or qq_rcut > L_new / 2:
        return False, box, energy_old, None        # minimum image would break: reject outright
    tot = ctx.volume_trial(L_new, alpha / L_new, lj_rcut, qq_rcut)
    energy_new = tot["energy"]
    beta = 1.0 / temperature
    arg = -beta * (pressure * (vol_new - vol_old) - n_mol * math.log(vol_new / vol_old) / beta
                   + (energy_new - energy_old))
    test = math.exp(min(arg, 700.0))
    if rng.random() < test:
        ctx.volume_accept()
        return True, L_new, energy_new, tot
    ctx.volume_reject()
    return False, box, energy_old, None


# ---- Lennard-Jones tail corrections (Ewald/energy.jl:514-614) --------------------------------------
def ener_corr(eps, sig, r_cut, box, b):
    """energy.jl:564-612 `ener_corr`: 8 pi / (3 V) * sum_ij b_i b_j eps_ij sig_ij^3
    ((1/3) (sig_ij/r_c)^9 - (sig_ij/r_c)^3), b = number of atoms of each type."""
    eps, sig, b = (np.asarray(x, dtype=float) for x in (eps, sig, b))
    vol = float(box) ** 3
    sig3 = sig ** 3
    sigor3 = sig3 / float(r_cut) ** 3
    coru = (b[:, None] * b[None, :] * eps * sig3 * ((1.0 / 3.0) * sigor3 ** 3 - sigor3)).sum()
    return 8.0 * np.pi / (3.0 * vol) * coru


def press_corr(eps, sig, r_cut, box, b):
    """energy.jl:514-562 `press_corr`: 16 pi / (3 V^2) * sum_ij b_i b_j eps_ij sig_ij^3
    ((2/3) (sig_ij/r_c)^9 - (sig_ij/r_c)^3)."""
    eps, sig, b = (np.asarray(x, dtype=float) for x in (eps, sig, b))
    vol = float(box) ** 3
    sig3 = sig ** 3
    sigor3 = sig3 / float(r_cut) ** 3
    corp = (b[:, None] * b[None, :] * eps * sig3 * ((2.0 / 3.0) * sigor3 ** 3 - sigor3)).sum()
    return 16.0 * np.pi / (3.0 * vol * vol) * corp
