"""Shared helpers for the tests: load the committed fixtures and build matching inputs for the
oracle (oracle.oracle.System / Ewald) and for the product (numpy arrays for device.Context)."""
import json
import os

import numpy as np

from metropolismontecarlo_amd import io as mio

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

# NIST SPC/E reference calculations for the four sample configurations (10 A cutoff,
# alpha = 5.6/L, kmax = 5, k^2 < 27); energies / k_B in K.  BASELINE.md section 2.
NIST = {
    1: dict(n=100, L=20.0, disp=9.95387e4, lrc=-8.23715e2, real=-5.58889e5, fourier=6.27009e3,
            self=-2.84469e6, intra=2.80999e6),
    2: dict(n=200, L=20.0, disp=1.93712e5, real=-1.19295e6, fourier=6.03495e3, self=-5.68938e6,
            intra=5.61998e6),
    3: dict(n=300, L=20.0, disp=3.54344e5, real=-1.96297e6, fourier=5.24461e3, self=-8.53407e6,
            intra=8.42998e6),
    4: dict(n=750, L=30.0, disp=4.48593e5, real=-3.57226e6, fourier=7.58785e3, self=-1.42235e7,
            intra=1.41483e7),
}

_golden = None


def nist_arrays(k, variant="reference"):
    return mio.load_nist_fixture(k, variant)


def golden(k, variant="reference"):
    global _golden
    if _golden is None:
        with open(os.path.join(GOLDEN, "golden_oracle.json")) as fh:
            _golden = json.load(fh)
    return _golden[f"config{k}_{variant}"]


def oracle_system(a):
    from oracle import oracle as orc
    return orc.System(a["com"], a["first_atom"], a["last_atom"], a["coords"], a["atype"],
                      a["charge"], a["eps"], a["sig"], a["box"])


def device_context(a, ewald=True):
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Context
    ctx = Context()
    ctx.upload_system(a["com"], a["first_atom"], a["last_atom"], a["coords"], a["atype"],
                      a["charge"], a["eps"], a["sig"], a["box"])
    if ewald:
        ctx.prepare_ewald(5.6 / a["box"], 5, 27, a["box"], structs.factor)
    return ctx


def rel(a, b, floor=0.0):
    return abs(a - b) / max(abs(b), floor, 1e-300)


def random_system(n_mol, box, seed, na_choices=(3,), n_types=2, min_sep=2.2):
    """A random rigid-molecule system with optionally ragged molecules (na in na_choices), random
    charges (neutral overall), and an LJ table with some zero entries."""
    rng = np.random.default_rng(seed)
    # place COMs on a jittered lattice so that nothing overlaps unless asked for
    nc = int(np.ceil(n_mol ** (1 / 3)))
    d = box / nc
    sites = np.array([(i, j, k) for i in range(nc) for j in range(nc) for k in range(nc)])[:n_mol]
    com = (sites + 0.5) * d + (rng.random((n_mol, 3)) - 0.5) * max(d - min_sep, 0.0) * 0.5
    na = rng.choice(na_choices, size=n_mol)
    first = np.concatenate([[1], 1 + np.cumsum(na)[:-1]]).astype(np.int64)
    last = (first + na - 1).astype(np.int64)
    n_atoms = int(na.sum())
    coords = np.empty((n_atoms, 3))
    for m in range(n_mol):
        off = rng.normal(size=(na[m], 3)) * 0.45
        off -= off.mean(0)
        coords[first[m] - 1:last[m]] = com[m] + off
    atype = rng.integers(1, n_types + 1, size=n_atoms).astype(np.int64)
    charge = rng.normal(size=n_atoms) * 0.5
    charge -= charge.mean()
    e = rng.random(n_types) * 100.0
    e[-1] = 0.0  # a type without LJ, like the SPC/E hydrogens
    s = 2.5 + rng.random(n_types)
    eps = np.sqrt(e[:, None] * e[None, :])
    sig = (s[:, None] + s[None, :]) / 2
    return dict(com=com, first_atom=first, last_atom=last, coords=coords, atype=atype,
                charge=charge, eps=eps, sig=sig, box=float(box))
