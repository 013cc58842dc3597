"""Observables built on the device state: normalisation of the radial
distribution histogram exactly as Ewald/gr.jl:92-104 writes it."""
import numpy as np


def normalize_rdf(hist, npart, side, nstep):
    """gr.jl:92-104: phi = npart / side^3, norm = 2 pi dr phi nstep npart,
    g(r_i) = hist[i] / norm / (r_i^2 + dr^2/12) at r_i = (i - 1/2) dr, i = 1..numbins.
    `hist` has numbins + 1 entries (index 0 unused, as in gr.jl).  Returns (r, g)."""
    hist = np.asarray(hist, dtype=float)
    numbins = hist.shape[0] - 1
    dr = (side / 2.0) / numbins
    phi = npart / side ** 3
    norm = 2.0 * np.pi * dr * phi * nstep * npart
    i = np.arange(1, numbins + 1)
    rrr = (i - 0.5) * dr
    return rrr, hist[1:] / norm / (rrr * rrr + dr * dr / 12.0)
