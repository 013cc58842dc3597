"""Independent numpy/scipy statement of the hot-path maths -- a SECOND opinion on the C oracle.

TEST INFRASTRUCTURE ONLY (see mmc_oracle.h).  Deliberately written differently from
mmc_oracle.c: vectorised, minimum image by rounding instead of comparisons, phases by direct
exp(i k.r) instead of the power recurrence, scipy.special.erfc instead of libm.  Agreement to
~1e-12 therefore checks the formulae (cutoff logic, factors, index conventions), not the rounding.
Reference lines: Ewald/energy.jl:209-290, Ewald/ewalds.jl:293-376, :538-604, :718-826.
"""
import numpy as np
from scipy.special import erfc


def _mi(d, L):
    return d - L * np.round(d / L)


def _neighbours(i, a, gate):
    com = a["com"]
    d = _mi(com - com[i - 1], a["box"])
    r2 = (d * d).sum(1)
    mask = r2 < gate * gate
    mask[i - 1] = False
    return np.nonzero(mask)[0], d


def lj_poly_du(i, a, r_cut):
    L = a["box"]
    js, dcom = _neighbours(i, a, r_cut)
    fa, la = a["first_atom"], a["last_atom"]
    pot = vir = 0.0
    ia = np.arange(fa[i - 1] - 1, la[i - 1])
    for j in js:
        jb = np.arange(fa[j] - 1, la[j])
        rab = _mi(a["coords"][jb][None, :, :] - a["coords"][ia][:, None, :], L)
        r2 = (rab * rab).sum(2)
        e = a["eps"][a["atype"][ia][:, None] - 1, a["atype"][jb][None, :] - 1]
        s = a["sig"][a["atype"][ia][:, None] - 1, a["atype"][jb][None, :] - 1]
        on = (r2 < r_cut ** 2 + 100) & (e > 0.001)
        s2 = np.where(on, s * s / r2, 0.0)
        s6 = s2 ** 3
        s12 = s6 ** 2
        pot += (e * (s12 - s6))[on].sum()
        virab = e * (2 * s12 - s6) * s2
        vir += ((rab * virab[:, :, None]).sum((0, 1)) * dcom[j])[...].sum() if on.any() else 0.0
    return 4 * pot, 24 * vir / 3.0


def ewald_real(i, a, kappa, r_cut, ovr):
    L = a["box"]
    js, _ = _neighbours(i, a, r_cut)
    fa, la = a["first_atom"], a["last_atom"]
    ia = np.arange(fa[i - 1] - 1, la[i - 1])
    pot = 0.0
    for j in js:
        jb = np.arange(fa[j] - 1, la[j])
        rab = _mi(a["coords"][jb][None, :, :] - a["coords"][ia][:, None, :], L)
        r2 = (rab * rab).sum(2)
        qq = a["charge"][ia][:, None] * a["charge"][jb][None, :]
        if ((r2 < ovr) & (qq < 0)).any():
            return 0.0, True
        r = np.sqrt(r2)
        on = r2 < r_cut ** 2 + 100
        pot += (qq * erfc(kappa * r) / r)[on].sum()
    return pot, False


def structure_factor(kxyz, coords, charge, box):
    phase = 2 * np.pi / box * (coords @ kxyz.T.astype(float))  # (n_atoms, n_k)
    return (charge[:, None] * np.exp(1j * phase)).sum(0)


def recip_long(kxyz, cfac, coords, charge, box):
    S = structure_factor(kxyz, np.asarray(coords), np.asarray(charge), box)
    return float((cfac * np.abs(S) ** 2).sum()), S


def recip_move_delta(kxyz, cfac, S_old, r_old, r_new, q, box):
    dS = structure_factor(kxyz, r_new, q, box) - structure_factor(kxyz, r_old, q, box)
    return float((cfac * (np.abs(S_old + dS) ** 2 - np.abs(S_old) ** 2)).sum())


def make_rdf_hist(sites, side, numbins):
    """Ewald/gr.jl:60-91 (`makeRDF`, the pair loop), literally: all pairs i < j, the file's own
    minimum image (strict < -side/2 -> + side, > side/2 -> - side), bin = ceil(r / dr) with
    dr = side / 2 / numbins, counted when bin <= numbins.  hist[0 .. numbins]."""
    sites = np.asarray(sites, dtype=np.float64)
    sideh = side / 2.0
    dr = sideh / numbins
    hist = np.zeros(numbins + 1, dtype=np.uint64)
    for i in range(sites.shape[0] - 1):
        d = sites[i] - sites[i + 1:]
        d = np.where(d < -sideh, d + side, d)
        d = np.where(d > sideh, d - side, d)
        rij = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        b = np.ceil(rij / dr)
        b = b[b <= numbins].astype(np.int64)
        np.add.at(hist, b, 1)
    return hist
