"""Host-side counterparts of the reference types that cross the hot-path boundary.

Field names are the reference's (Ewald/ewalds.jl:9-19, Ewald/auxillary.jl:37-91,
Ewald/structs.jl:283-347).  Julia identifiers that are not ASCII keep an ASCII alias:
``Tables.eps_ij`` is ``Tables.ϵᵢⱼ``, ``Properties2.rho`` is ``ρ``.  Arrays are numpy: a
``Vector{SVector{3,Float64}}`` is an (n, 3) float64 array, index arrays keep Julia's 1-based
values.
"""
import math
from dataclasses import dataclass, field

import numpy as np


# Ewald/constants.jl:24-28 (same order of operations -> same double)
def _factor():
    kb1 = 1.3806488e-23
    eps01 = 8.854187817e-12
    eps01 *= 1e-10
    e1 = 1.602176565e-19
    return e1 ** 2 / eps01 / 4 / math.pi / kb1


factor = _factor()
R = 8.3144621e-3  # kJ mol^-1 K^-1, Ewald/constants.jl:11


class StructArray:
    """Minimal stand-in for StructArrays.StructArray: one numpy column per field, the way the
    hot path reads it (`moa.COM`, `moa.firstAtom`, `soa.coords`, `soa.charge`, `soa.atype`)."""

    def __init__(self, **columns):
        n = None
        for k, v in columns.items():
            v = np.asarray(v)
            n = len(v) if n is None else n
            if len(v) != n:
                raise ValueError(f"column {k} has length {len(v)}, expected {n}")
            setattr(self, k, v)
        self._names = list(columns)
        self._n = n or 0

    def __len__(self):
        return self._n


def make_moa(COM, firstAtom, lastAtom, molType=None):
    """moa: ParticleAtomKMC/Molecule columns used by the path (Ewald/structs.jl:314-321)."""
    COM = np.ascontiguousarray(COM, dtype=np.float64).reshape(-1, 3)
    n = COM.shape[0]
    return StructArray(COM=COM, firstAtom=np.ascontiguousarray(firstAtom, dtype=np.int64),
                       lastAtom=np.ascontiguousarray(lastAtom, dtype=np.int64),
                       molType=np.ones(n, dtype=np.int64) if molType is None else
                       np.asarray(molType, dtype=np.int64))


def make_soa(coords, atype, charge, molNum=None):
    """soa: atom columns used by the path (Ewald/structs.jl:283-290)."""
    coords = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, 3)
    n = coords.shape[0]
    return StructArray(coords=coords, atype=np.ascontiguousarray(atype, dtype=np.int64),
                       charge=np.ascontiguousarray(charge, dtype=np.float64),
                       molNum=np.zeros(n, dtype=np.int64) if molNum is None else
                       np.asarray(molNum, dtype=np.int64))


class Tables:
    """Ewald/structs.jl:337-347: eps_ij = sqrt(eps_i eps_j), sig_ij = (sig_i + sig_j)/2."""

    def __init__(self, a, b):
        a = np.asarray(a, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        self.eps_ij = np.sqrt(a[:, None] * a[None, :])
        self.sig_ij = (b[:, None] + b[None, :]) / 2

    ϵᵢⱼ = property(lambda s: s.eps_ij, lambda s, v: setattr(s, "eps_ij", np.asarray(v, float)))
    σᵢⱼ = property(lambda s: s.sig_ij, lambda s, v: setattr(s, "sig_ij", np.asarray(v, float)))


class EWALD:
    """mutable struct EWALD{I} (Ewald/ewalds.jl:9-19)."""

    def __init__(self, kappa, nk, k_sq_max, NKVECS, kxyz, cfac, sumQExpOld, sumQExpNew, factor):
        self.kappa = float(kappa)
        self.nk = int(nk)
        self.k_sq_max = int(k_sq_max)
        self.NKVECS = int(NKVECS)
        self.kxyz = np.asarray(kxyz, dtype=np.int32)
        self.cfac = np.asarray(cfac, dtype=np.float64)
        self.sumQExpOld = np.asarray(sumQExpOld, dtype=np.complex128)
        self.sumQExpNew = np.asarray(sumQExpNew, dtype=np.complex128)
        self.factor = float(factor)


@dataclass
class Properties:
    """Ewald/auxillary.jl:37-45"""
    energy: float = 0.0
    virial: float = 0.0
    coulomb: float = 0.0
    recip: float = 0.0
    recipOld: float = 0.0
    old_e: float = 0.0
    old_v: float = 0.0


@dataclass
class Moves:
    """Ewald/auxillary.jl:48-55"""
    naccepp: int = 0
    naccept: int = 0
    attempp: int = 0
    attempt: int = 0
    set_value: float = 0.5
    d_max: float = 0.0


@dataclass
class Properties2:
    """Ewald/auxillary.jl:78-91 (rho = the reference's field `ρ`, dphi_max = `dϕ_max`)."""
    temperature: float = 298.15
    rho: float = 0.0
    pressure: float = 0.0
    dr_max: float = 0.0
    dphi_max: float = 0.0
    move_accept: float = 0.3
    numTranAccepted: int = 0
    totalStepsTaken: int = 0
    quat: list = field(default_factory=list)
    LJ_rcut: float = 10.0
    qq_rcut: float = 10.0
    box: float = 0.0


@dataclass
class Requirements:
    """Legacy argument bundle (Ewald/auxillary.jl:59-75)."""
    rm: np.ndarray
    ra: np.ndarray
    nMols: int
    nAtoms: int
    nCharges: int
    thisMol_theseAtoms: np.ndarray  # (n_mol, 2) 1-based inclusive
    molNames: list
    molTypes: list
    atomNames: list
    atomTypes: np.ndarray
    table: Tables
    box: float
    r_cut: float
