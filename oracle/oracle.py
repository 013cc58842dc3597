"""ctypes binding of the CPU oracle (oracle/mmc_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never from ``metropolismontecarlo_amd``.  Array
conventions are the Julia ones (1-based inclusive atom ranges, 1-based atom types, column-major
LJ tables); see mmc_oracle.h for the reference file:line each function follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmmc_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "mmc_oracle.c")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src),
                                             os.path.getmtime(os.path.join(_HERE, "mmc_oracle.h")))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmmc_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)


class Totals(C.Structure):
    _fields_ = [("energy", C.c_double), ("virial", C.c_double), ("coulomb", C.c_double),
                ("lj", C.c_double), ("real", C.c_double), ("recip", C.c_double),
                ("self", C.c_double), ("n_overlap", C.c_int32)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_factor.restype = C.c_double
        _lib.orc_vector1D.restype = C.c_double
        _lib.orc_vector1D.argtypes = [C.c_double] * 3
        _lib.orc_prepare_ewald.restype = C.c_int64
        _lib.orc_ewald_real_atomcut.restype = C.c_double
        _lib.orc_recip_long.restype = C.c_double
        _lib.orc_ewald_self.restype = C.c_double
        _lib.orc_recip_move.restype = C.c_int32
        _lib.orc_coulomb_real.restype = C.c_int32
        _lib.orc_trial_move.restype = C.c_int32
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class System:
    """The hot path's inputs as flat arrays (the moa/soa StructArray fields + Tables)."""

    def __init__(self, com, first_atom, last_atom, coords, atype, charge, eps, sig, box):
        self.com = _f64(com).reshape(-1, 3).copy()
        self.first_atom = _i64(first_atom).copy()
        self.last_atom = _i64(last_atom).copy()
        self.coords = _f64(coords).reshape(-1, 3).copy()
        self.atype = _i64(atype).copy()
        self.charge = _f64(charge).copy()
        # Julia Matrix is column-major; tables are symmetric but keep the layout honest
        self.eps = np.asfortranarray(eps, dtype=np.float64)
        self.sig = np.asfortranarray(sig, dtype=np.float64)
        self.n_types = self.eps.shape[0]
        self.box = float(box)

    @property
    def n_mol(self):
        return self.com.shape[0]

    @property
    def n_atoms(self):
        return self.coords.shape[0]

    def copy(self):
        return System(self.com, self.first_atom, self.last_atom, self.coords, self.atype,
                      self.charge, self.eps, self.sig, self.box)


class Ewald:
    """EWALD struct (Ewald/ewalds.jl:9-19) after PrepareEwaldVariables (:45-103)."""

    def __init__(self, kappa, nk, k_sq_max, box, factor=None):
        L = lib()
        self.kappa, self.nk, self.k_sq_max = float(kappa), int(nk), int(k_sq_max)
        self.factor = L.orc_factor() if factor is None else float(factor)
        n = L.orc_prepare_ewald(C.c_double(kappa), C.c_int64(nk), C.c_int64(k_sq_max),
                                C.c_double(box), None, None)
        if n < 0:
            raise AssertionError("k_sq_max == 27")
        self.NKVECS = int(n)
        self.kxyz = np.zeros((n, 3), dtype=np.int32)
        self.cfac = np.zeros(n, dtype=np.float64)
        L.orc_prepare_ewald(C.c_double(kappa), C.c_int64(nk), C.c_int64(k_sq_max),
                            C.c_double(box), self.kxyz.ctypes.data_as(_i32p), _d(self.cfac))
        self.sumQExpOld = np.zeros(n, dtype=np.complex128)
        self.sumQExpNew = np.zeros(n, dtype=np.complex128)


def factor():
    return lib().orc_factor()


def vector1D(c1, c2, box):
    return lib().orc_vector1D(c1, c2, box)


def lj_poly_du(i, s, r_cut):
    pot, vir = C.c_double(), C.c_double()
    lib().orc_lj_poly_du(C.c_int64(i), C.c_int64(s.n_mol), _d(s.com), _i(s.first_atom),
                         _i(s.last_atom), _d(s.coords), _i(s.atype), C.c_int64(s.n_types),
                         _d(s.eps), _d(s.sig), C.c_double(r_cut), C.c_double(s.box),
                         C.byref(pot), C.byref(vir))
    return pot.value, vir.value


def ewald_real(i, s, kappa, r_cut, ovr=0.5):
    pot, ov = C.c_double(), C.c_int32()
    lib().orc_ewald_real(C.c_int64(i), C.c_int64(s.n_mol), _d(s.com), _i(s.first_atom),
                         _i(s.last_atom), _d(s.coords), _d(s.charge), C.c_double(kappa),
                         C.c_double(r_cut), C.c_double(s.box), C.c_double(ovr), C.byref(pot),
                         C.byref(ov))
    return pot.value, bool(ov.value)


def ewald_short(i, s, ew, qq_rcut):
    e, v, ov = C.c_double(), C.c_double(), C.c_int32()
    lib().orc_ewald_short(C.c_int64(i), C.c_int64(s.n_mol), _d(s.com), _i(s.first_atom),
                          _i(s.last_atom), _d(s.coords), _d(s.charge), C.c_double(ew.kappa),
                          C.c_double(qq_rcut), C.c_double(s.box), C.c_double(ew.factor),
                          C.byref(e), C.byref(v), C.byref(ov))
    return e.value, v.value, bool(ov.value)


def coulomb_real(i, s, r_cut):
    pot, ov = C.c_double(), C.c_int32()
    st = lib().orc_coulomb_real(C.c_int64(i), C.c_int64(s.n_mol), _d(s.com), _i(s.first_atom),
                                _i(s.last_atom), _d(s.coords), _d(s.charge), C.c_double(r_cut),
                                C.c_double(s.box), C.byref(pot), C.byref(ov))
    if st != 0:
        raise AssertionError("r_cut == 10.0")
    return pot.value, bool(ov.value)


def ewald_real_atomcut(i, s, kappa, r_cut):
    return lib().orc_ewald_real_atomcut(C.c_int64(i), _i(s.first_atom), _i(s.last_atom),
                                        C.c_int64(s.n_atoms), _d(s.coords), _d(s.charge),
                                        C.c_double(kappa), C.c_double(r_cut), C.c_double(s.box))


def _cview(z):
    assert z.dtype == np.complex128 and z.flags.c_contiguous
    return z.view(np.float64).ctypes.data_as(_dp)


def recip_long(ew, coords, charge, box):
    coords, charge = _f64(coords), _f64(charge)
    return lib().orc_recip_long(C.c_int64(ew.nk), C.c_int64(ew.NKVECS),
                                ew.kxyz.ctypes.data_as(_i32p), _d(ew.cfac),
                                C.c_int64(charge.shape[0]), _d(coords), _d(charge),
                                C.c_double(box), _cview(ew.sumQExpOld), _cview(ew.sumQExpNew))


def recip_move(box, ew, r_old, r_new, q):
    r_old, r_new, q = _f64(r_old), _f64(r_new), _f64(q)
    de = C.c_double()
    st = lib().orc_recip_move(C.c_double(box), C.c_int64(ew.nk), C.c_int64(ew.k_sq_max),
                              C.c_int64(ew.NKVECS), ew.kxyz.ctypes.data_as(_i32p), _d(ew.cfac),
                              _cview(ew.sumQExpOld), _cview(ew.sumQExpNew), _d(r_old), _d(r_new),
                              _d(q), C.c_int64(q.shape[0]), C.c_double(ew.factor), C.byref(de))
    if st != 0:
        raise AssertionError("n == 3 && k_sq_max == 27 && nk == 5")
    return de.value


def ewald_self(ew, charge):
    charge = _f64(charge)
    return lib().orc_ewald_self(C.c_double(ew.kappa), C.c_double(ew.factor),
                                C.c_int64(charge.shape[0]), _d(charge))


def potential_ewald(s, ew, lj_rcut, qq_rcut):
    t = Totals()
    lib().orc_potential_ewald(C.c_int64(s.n_mol), C.c_int64(s.n_atoms), _d(s.com),
                              _i(s.first_atom), _i(s.last_atom), _d(s.coords), _i(s.atype),
                              _d(s.charge), C.c_int64(s.n_types), _d(s.eps), _d(s.sig),
                              C.c_double(lj_rcut), C.c_double(qq_rcut), C.c_double(s.box),
                              C.c_double(ew.kappa), C.c_int64(ew.nk), C.c_int64(ew.NKVECS),
                              ew.kxyz.ctypes.data_as(_i32p), _d(ew.cfac), C.c_double(ew.factor),
                              _cview(ew.sumQExpOld), _cview(ew.sumQExpNew), C.byref(t))
    return t.asdict()


def potential_wolf(s, ew, lj_rcut, qq_rcut, literal_prefactor=True):
    t = Totals()
    lib().orc_potential_wolf(C.c_int64(s.n_mol), C.c_int64(s.n_atoms), _d(s.com),
                             _i(s.first_atom), _i(s.last_atom), _d(s.coords), _i(s.atype),
                             _d(s.charge), C.c_int64(s.n_types), _d(s.eps), _d(s.sig),
                             C.c_double(lj_rcut), C.c_double(qq_rcut), C.c_double(s.box),
                             C.c_double(ew.kappa), C.c_double(ew.factor),
                             C.c_int32(1 if literal_prefactor else 0), C.byref(t))
    return t.asdict()


def lj_du_monatomic(i, r, eps, sig, r_cut, box):
    r, eps, sig = _f64(r), _f64(eps), _f64(sig)
    pot, vir = C.c_double(), C.c_double()
    lib().orc_lj_du_monatomic(C.c_int64(i), C.c_int64(eps.shape[0]), _d(r), _d(eps), _d(sig),
                              C.c_double(r_cut), C.c_double(box), C.byref(pot), C.byref(vir))
    return pot.value, vir.value


def potential_monatomic(r, eps, sig, r_cut, box):
    r, eps, sig = _f64(r), _f64(eps), _f64(sig)
    e, v = C.c_double(), C.c_double()
    lib().orc_potential_monatomic(C.c_int64(eps.shape[0]), _d(r), _d(eps), _d(sig),
                                  C.c_double(r_cut), C.c_double(box), C.byref(e), C.byref(v))
    return e.value, v.value


def trial_move(i, s, ew, lj_rcut, qq_rcut, com_new, atoms_new):
    """Hot-path calls of one Loop() iteration (Ewald/main.jl:491-629).  Leaves ``s`` in the OLD
    state and ``ew.sumQExpNew`` mutated as RecipMove leaves it.  Returns (d[4], overlap)."""
    com_new, atoms_new = _f64(com_new), _f64(atoms_new)
    d = np.zeros(4)
    ov = C.c_int32()
    st = lib().orc_trial_move(C.c_int64(i), C.c_int64(s.n_mol), _d(s.com), _i(s.first_atom),
                              _i(s.last_atom), _d(s.coords), _i(s.atype), _d(s.charge),
                              C.c_int64(s.n_types), _d(s.eps), _d(s.sig), C.c_double(lj_rcut),
                              C.c_double(qq_rcut), C.c_double(s.box), C.c_double(ew.kappa),
                              C.c_int64(ew.nk), C.c_int64(ew.k_sq_max), C.c_int64(ew.NKVECS),
                              ew.kxyz.ctypes.data_as(_i32p), _d(ew.cfac), C.c_double(ew.factor),
                              _cview(ew.sumQExpOld), _cview(ew.sumQExpNew), _d(com_new),
                              _d(atoms_new), _d(d), C.byref(ov))
    if st != 0:
        raise AssertionError("RecipMove asserts (n == 3, k_sq_max == 27, nk == 5)")
    return d, bool(ov.value)


def bench_trial_moves(s, ew, lj_rcut, qq_rcut, dr_max, seed, n_threads, seconds):
    """Timing helper (orc_bench_trial_moves): a C loop of trial moves in n_threads independent
    copies of the system.  Returns (total trial moves, wall seconds of the slowest thread)."""
    L = lib()
    L.orc_bench_trial_moves.restype = C.c_int64
    el = C.c_double()
    n = L.orc_bench_trial_moves(
        C.c_int64(s.n_mol), C.c_int64(s.coords.shape[0]), _d(s.com), _i(s.first_atom),
        _i(s.last_atom), _d(s.coords), _i(s.atype), _d(s.charge), C.c_int64(s.n_types), _d(s.eps),
        _d(s.sig), C.c_double(lj_rcut), C.c_double(qq_rcut), C.c_double(s.box),
        C.c_double(ew.kappa), C.c_int64(ew.nk), C.c_int64(ew.k_sq_max), C.c_int64(ew.NKVECS),
        ew.kxyz.ctypes.data_as(_i32p), _d(ew.cfac), C.c_double(ew.factor), _cview(ew.sumQExpOld),
        C.c_double(dr_max), C.c_uint64(seed), C.c_int32(n_threads), C.c_double(seconds),
        C.byref(el))
    return int(n), float(el.value)
