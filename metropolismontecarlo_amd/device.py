"""Thin object wrappers over the C ABI (include/mmc_hip.h): ``Context`` (one system, the
reference's per-call surface) and ``Batch`` (R replicas, one launch per step).

numpy arrays in, numpy arrays / floats out.  Index conventions are the reference's: molecule
indices, atom ranges and atom types are 1-based.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CHAIN_DTYPE, TOTALS_DTYPE, Move, MoveResult, RunParams, RunStats, Totals, check

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_i64p)


REFERENCE_IDEAL_TERM = 4.60453   # the ideal-gas term Loop() hard-codes in its block line (main.jl:677)


def block_line(chain, block, n_mol, box, ideal_term=REFERENCE_IDEAL_TERM):
    """Loop()'s status line of one block (Ewald/main.jl:667-679) from one row of the chains array
    (mmc_chain_block_line).  ideal_term: the reference's literal 4.60453, or rho * T for
    auxillary.jl:121-123's Pressure(vir, rho, T, vol)."""
    row = np.ascontiguousarray(chain).reshape(1)
    if row.dtype != CHAIN_DTYPE:
        raise ValueError("chain must be a row of the array returned by Batch.new_chains()")
    buf = C.create_string_buffer(512)
    check(_lib.lib().mmc_chain_block_line(row.ctypes.data_as(C.c_void_p), int(block), int(n_mol),
                                          float(box), float(ideal_term), buf, 512))
    return buf.value.decode("utf-8")


def device_count():
    n = C.c_int32()
    check(_lib.lib().mmc_device_count(C.byref(n)))
    return n.value


class Context:
    """mmc_ctx: one system (one Markov chain) on one GPU."""

    def __init__(self, device=0, stream=None):
        self._L = _lib.lib()
        h = C.c_void_p()
        check(self._L.mmc_ctx_create(device, C.c_void_p(stream or 0), C.byref(h)))
        self._h = h
        self.n_mol = self.n_atoms = 0
        self.nkvecs = 0
        self.factor = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.mmc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- state ---------------------------------------------------------------------------------
    def upload_system(self, com, first_atom, last_atom, coords, atype, charge, eps, sig, box):
        com, coords, charge = _f64(com).reshape(-1, 3), _f64(coords).reshape(-1, 3), _f64(charge)
        fa, la, at = _i64(first_atom), _i64(last_atom), _i64(atype)
        eps = np.asfortranarray(eps, dtype=np.float64)
        sig = np.asfortranarray(sig, dtype=np.float64)
        nt = eps.shape[0]
        if eps.shape != (nt, nt) or sig.shape != (nt, nt):
            raise ValueError("eps/sig must be square n_types x n_types")
        if not (len(fa) == len(la) == com.shape[0]) or not (len(at) == len(charge) == coords.shape[0]):
            raise ValueError("inconsistent array lengths")
        ef, sf = eps.ravel(order="F").copy(), sig.ravel(order="F").copy()
        check(self._L.mmc_upload_system(self._h, com.shape[0], coords.shape[0], _d(com), _i(fa),
                                        _i(la), _d(coords), _i(at), _d(charge), nt, _d(ef),
                                        _d(sf), float(box)))
        self.n_mol, self.n_atoms = com.shape[0], coords.shape[0]
        self.box = float(box)

    def set_molecule(self, i, com, atoms):
        com, atoms = _f64(com).ravel(), _f64(atoms).ravel()
        check(self._L.mmc_set_molecule(self._h, int(i), _d(com), _d(atoms)))

    def update_system(self, com, coords):
        com, coords = _f64(com).reshape(-1, 3), _f64(coords).reshape(-1, 3)
        if com.shape[0] != self.n_mol or coords.shape[0] != self.n_atoms:
            raise ValueError("update_system: topology changed, create a new Context")
        check(self._L.mmc_update_system(self._h, _d(com), _d(coords)))

    def download_system(self):
        com = np.empty((self.n_mol, 3))
        coords = np.empty((self.n_atoms, 3))
        check(self._L.mmc_download_system(self._h, _d(com), _d(coords)))
        return com, coords

    def synchronize(self):
        check(self._L.mmc_ctx_synchronize(self._h))

    def volume_change(self, new_box, new_kappa):
        """Device part of an NPT volume move (volumeChange.jl:59-80): rescale + rebuild tables."""
        check(self._L.mmc_volume_change(self._h, float(new_box), float(new_kappa)))
        self.box = float(new_box)

    def volume_trial(self, new_box, new_kappa, lj_rcut, qq_rcut):
        """mmc_volume_trial: snapshot on the device, rescale, new tables, total energy at the new
        volume.  Follow with volume_accept() or volume_reject()."""
        t = Totals()
        check(self._L.mmc_volume_trial(self._h, float(new_box), float(new_kappa), float(lj_rcut),
                                       float(qq_rcut), C.byref(t)))
        self._box_before_trial = self.box
        self.box = float(new_box)
        return t.asdict()

    def volume_accept(self):
        check(self._L.mmc_volume_accept(self._h))

    def volume_reject(self):
        check(self._L.mmc_volume_reject(self._h))
        self.box = self._box_before_trial

    # -- a6 ------------------------------------------------------------------------------------
    def prepare_ewald(self, kappa, nk, k_sq_max, box, factor):
        n = C.c_int64()
        check(self._L.mmc_prepare_ewald(self._h, float(kappa), int(nk), int(k_sq_max), float(box),
                                        float(factor), C.byref(n)))
        self.nkvecs = n.value
        self.factor = float(factor)
        return n.value

    def get_kvectors(self):
        kxyz = np.empty((self.nkvecs, 3), dtype=np.int32)
        cfac = np.empty(self.nkvecs)
        check(self._L.mmc_get_kvectors(self._h, kxyz.ctypes.data_as(_i32p), _d(cfac)))
        return kxyz, cfac

    def get_sumqexp(self):
        so = np.empty(self.nkvecs, dtype=np.complex128)
        sn = np.empty(self.nkvecs, dtype=np.complex128)
        check(self._L.mmc_get_sumqexp(self._h, so.view(np.float64).ctypes.data_as(_dp),
                                      sn.view(np.float64).ctypes.data_as(_dp)))
        return so, sn

    def set_sumqexp(self, sum_old=None, sum_new=None):
        so = None if sum_old is None else np.ascontiguousarray(sum_old, dtype=np.complex128)
        sn = None if sum_new is None else np.ascontiguousarray(sum_new, dtype=np.complex128)
        check(self._L.mmc_set_sumqexp(
            self._h, None if so is None else so.view(np.float64).ctypes.data_as(_dp),
            None if sn is None else sn.view(np.float64).ctypes.data_as(_dp)))

    # -- a2..a13 -------------------------------------------------------------------------------
    def lj_poly_du(self, i, r_cut):
        p, v = C.c_double(), C.c_double()
        check(self._L.mmc_lj_poly_du(self._h, int(i), float(r_cut), C.byref(p), C.byref(v)))
        return p.value, v.value

    def ewald_real(self, i, r_cut, ovr=0.5):
        p, o = C.c_double(), C.c_int32()
        check(self._L.mmc_ewald_real(self._h, int(i), float(r_cut), float(ovr), C.byref(p),
                                     C.byref(o)))
        return p.value, bool(o.value)

    def ewald_short(self, i, qq_rcut):
        e, v, o = C.c_double(), C.c_double(), C.c_int32()
        check(self._L.mmc_ewald_short(self._h, int(i), float(qq_rcut), C.byref(e), C.byref(v),
                                      C.byref(o)))
        return e.value, v.value, bool(o.value)

    def coulomb_real(self, i, r_cut):
        p, o = C.c_double(), C.c_int32()
        check(self._L.mmc_coulomb_real(self._h, int(i), float(r_cut), C.byref(p), C.byref(o)))
        return p.value, bool(o.value)

    def recip_long(self):
        e = C.c_double()
        check(self._L.mmc_recip_long(self._h, C.byref(e)))
        return e.value

    def recip_move(self, r_old, r_new, q):
        r_old, r_new, q = _f64(r_old).ravel(), _f64(r_new).ravel(), _f64(q).ravel()
        e = C.c_double()
        check(self._L.mmc_recip_move(self._h, _d(r_old), _d(r_new), _d(q), q.shape[0],
                                     C.byref(e)))
        return e.value

    def recip_commit(self):
        check(self._L.mmc_recip_commit(self._h))

    def recip_rollback(self):
        check(self._L.mmc_recip_rollback(self._h))

    def ewald_self(self):
        e = C.c_double()
        check(self._L.mmc_ewald_self(self._h, C.byref(e)))
        return e.value

    def potential_ewald(self, lj_rcut, qq_rcut):
        t = Totals()
        check(self._L.mmc_potential_ewald(self._h, float(lj_rcut), float(qq_rcut), C.byref(t)))
        return t.asdict()

    def potential_wolf(self, lj_rcut, qq_rcut):
        t = Totals()
        check(self._L.mmc_potential_wolf(self._h, float(lj_rcut), float(qq_rcut), C.byref(t)))
        return t.asdict()

    def trial_move(self, i, com_new, atoms_new, lj_rcut, qq_rcut):
        com_new, atoms_new = _f64(com_new).ravel(), _f64(atoms_new).ravel()
        d = np.zeros(4)
        o = C.c_int32()
        check(self._L.mmc_trial_move(self._h, int(i), _d(com_new), _d(atoms_new), float(lj_rcut),
                                     float(qq_rcut), _d(d), C.byref(o)))
        return d, bool(o.value)

    # ---- single-precision tolerance study (not a reference interface) ----
    def study_f32_total(self, lj_rcut, qq_rcut, mixed=False):
        """{lj, lj_virial, real, recip, n_overlap}: the summed terms of potential() from fp32
        coordinates and arithmetic (mixed: fp64 accumulators)."""
        out = np.zeros(6)
        check(self._L.mmc_study_f32_total(self._h, float(lj_rcut), float(qq_rcut), int(bool(mixed)),
                                          _d(out)))
        return dict(lj=out[0], lj_virial=out[1], real=out[2], recip=out[3], n_overlap=int(out[4]))

    def study_f32_move(self, i, com_new, atoms_new, lj_rcut, qq_rcut, mixed=False):
        com_new, atoms_new = _f64(com_new).ravel(), _f64(atoms_new).ravel()
        d = np.zeros(3)
        o = C.c_int32()
        check(self._L.mmc_study_f32_move(self._h, int(i), _d(com_new), _d(atoms_new),
                                         float(lj_rcut), float(qq_rcut), int(bool(mixed)), _d(d),
                                         C.byref(o)))
        return d, bool(o.value)

    def accept_move(self):
        check(self._L.mmc_accept_move(self._h))

    def stats(self):
        """mmc_ctx_stats: counters of the context's engine (persistent kernel, cache, speculation)."""
        st = (C.c_int64 * 10)()
        check(self._L.mmc_ctx_stats(self._h, st))
        return dict(zip(("cmds", "launches", "retries", "cache_hits", "spec_hits", "spec_miss",
                         "launch_evals", "alive", "look_ahead_hits", "look_ahead_posted"), list(st)))

    def set_option(self, key, value):
        check(self._L.mmc_ctx_set_option(self._h, key.encode(), int(value)))

    def ping(self, n=1000):
        """Average round trip (us) of n empty commands through the running persistent kernel."""
        us = C.c_double()
        check(self._L.mmc_ctx_ping(self._h, int(n), C.byref(us)))
        return us.value

    def reject_move(self):
        check(self._L.mmc_reject_move(self._h))


class Batch:
    """mmc_batch: R replicas of one 3-atoms-per-molecule system on one GPU."""

    def __init__(self, n_replicas, com, coords, atype, charge, eps, sig, box, kappa, factor,
                 lj_rcut, qq_rcut, nk=5, k_sq_max=27, device=0, stream=None):
        self._L = _lib.lib()
        com, coords, charge = _f64(com).reshape(-1, 3), _f64(coords).reshape(-1, 3), _f64(charge)
        at = _i64(atype)
        eps = np.asfortranarray(eps, dtype=np.float64)
        sig = np.asfortranarray(sig, dtype=np.float64)
        nt = eps.shape[0]
        if coords.shape[0] != 3 * com.shape[0]:
            raise AssertionError("n == 3 atoms per molecule (ewalds.jl:740)")
        ef, sf = eps.ravel(order="F").copy(), sig.ravel(order="F").copy()
        h = C.c_void_p()
        check(self._L.mmc_batch_create(device, C.c_void_p(stream or 0), int(n_replicas),
                                       com.shape[0], _d(com), _d(coords), _i(at), _d(charge), nt,
                                       _d(ef), _d(sf), float(box), float(kappa), int(nk),
                                       int(k_sq_max), float(factor), float(lj_rcut),
                                       float(qq_rcut), C.byref(h)))
        self._h = h
        self.R, self.n_mol = int(n_replicas), com.shape[0]
        self.factor = float(factor)
        self.nkvecs = 337

    def close(self):
        if getattr(self, "_h", None):
            self._L.mmc_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_replica(self, r, com, coords):
        com, coords = _f64(com).reshape(-1, 3), _f64(coords).reshape(-1, 3)
        check(self._L.mmc_batch_set_replica(self._h, int(r), _d(com), _d(coords)))

    def get_replica(self, r):
        com = np.empty((self.n_mol, 3))
        coords = np.empty((3 * self.n_mol, 3))
        s = np.empty(self.nkvecs, dtype=np.complex128)
        check(self._L.mmc_batch_get_replica(self._h, int(r), _d(com), _d(coords),
                                            s.view(np.float64).ctypes.data_as(_dp)))
        return com, coords, s

    def recip_long(self):
        e = np.empty(self.R)
        check(self._L.mmc_batch_recip_long(self._h, _d(e)))
        return e

    def potential_ewald(self, as_array=False):
        """potential(..., "ewald") of every replica: a list of dicts, or with as_array=True one
        numpy record array (_lib.TOTALS_DTYPE) filled in place by the library."""
        if as_array:
            out = np.zeros(self.R, dtype=TOTALS_DTYPE)
            check(self._L.mmc_batch_potential_ewald(self._h, out.ctypes.data_as(C.c_void_p)))
            return out
        t = (Totals * self.R)()
        check(self._L.mmc_batch_potential_ewald(self._h, t))
        return [x.asdict() for x in t]

    def eval(self, mol, com_new, atoms_new, accept_prev=None):
        """mol: (R,) 1-based; com_new: (R,3); atoms_new: (R,3,3); accept_prev: (R,) bool."""
        mol = np.broadcast_to(np.asarray(mol, dtype=np.int64), (self.R,))
        com_new = _f64(com_new).reshape(self.R, 3)
        atoms_new = _f64(atoms_new).reshape(self.R, 9)
        acc = np.zeros(self.R, dtype=bool) if accept_prev is None else np.asarray(accept_prev)
        moves = (Move * self.R)()
        for r in range(self.R):
            moves[r].mol = int(mol[r])
            moves[r].accept_prev = int(bool(acc[r]))
            moves[r].com_new[:] = com_new[r].tolist()
            moves[r].atoms_new[:] = atoms_new[r].tolist()
        res = (MoveResult * self.R)()
        check(self._L.mmc_batch_eval(self._h, moves, res))
        out = np.array([[x.d_lj, x.d_real, x.d_recip, x.d_vir] for x in res])
        ov = np.array([bool(x.overlap) for x in res])
        return out, ov

    def set_parts(self, n_parts):
        check(self._L.mmc_batch_set_parts(self._h, int(n_parts)))

    def settle(self, accept):
        a = np.ascontiguousarray(accept, dtype=np.int32)
        check(self._L.mmc_batch_settle(self._h, a.ctypes.data_as(_i32p)))

    def set_option(self, key, value):
        check(self._L.mmc_batch_set_option(self._h, key.encode(), int(value)))

    def volume_change(self, new_box, new_kappa):
        check(self._L.mmc_batch_volume_change(self._h, float(new_box), float(new_kappa)))

    def volume_trial(self, new_box, new_kappa):
        """mmc_batch_volume_trial (a batch of ONE replica): device-side snapshot, rescale, new
        tables, total energy at the new volume.  Follow with volume_accept() or volume_reject()."""
        t = Totals()
        check(self._L.mmc_batch_volume_trial(self._h, float(new_box), float(new_kappa), C.byref(t)))
        return t.asdict()

    def volume_accept(self):
        check(self._L.mmc_batch_volume_accept(self._h))

    def volume_reject(self):
        check(self._L.mmc_batch_volume_reject(self._h))

    def run_npt(self, n_sweeps, temperature, pressure, vmax, dr_max, dphi_max, seed, energy,
                moves_per_sweep=0, alpha=5.6, n_parts=0, n_threads=1, replica0=0):
        """mmc_batch_run_npt: n_sweeps x { moves_per_sweep trial moves (0 = one per molecule), one
        volume move (Ewald/volumeChange.jl:59-147) }.  Returns (energy, run stats, npt stats)."""
        from ._lib import NptParams, NptStats
        p = RunParams(float(temperature), float(dr_max), float(dphi_max), int(seed), 0, 1,
                      int(n_parts), 0, int(n_threads), 0, 0, int(replica0))
        q = NptParams(float(pressure), float(vmax), float(alpha), int(n_sweeps), int(moves_per_sweep))
        st, ns = RunStats(), NptStats()
        e = np.array([float(energy)])
        check(self._L.mmc_batch_run_npt(self._h, C.byref(p), C.byref(q), _d(e), C.byref(st), C.byref(ns)))
        return float(e[0]), st.asdict(), ns.asdict()

    def qq_table(self, r2):
        r2 = _f64(r2).ravel()
        out = np.empty_like(r2)
        check(self._L.mmc_batch_qq_table(self._h, _d(r2), r2.shape[0], _d(out)))
        return out

    def run(self, n_steps, temperature, dr_max, dphi_max, seed, energies=None, n_groups=2,
            n_parts=0, time_kernels=False, n_threads=1, n_streams=0, replica0=0):
        """mmc_batch_run.  Chain r draws from the stream (seed, replica0 + r): `replica0` is the
        global index of this batch's first chain when an ensemble is spread over several GPUs."""
        p = RunParams(float(temperature), float(dr_max), float(dphi_max), int(seed), int(n_steps),
                      int(n_groups), int(n_parts), int(time_kernels), int(n_threads),
                      int(n_streams), 0, int(replica0))
        st = RunStats()
        e = np.zeros(self.R) if energies is None else _f64(energies).copy()
        check(self._L.mmc_batch_run(self._h, C.byref(p), _d(e), C.byref(st)))
        return e, st.asdict()

    def get_trace(self, n_steps):
        """(dU[R, n], flags[R, n]) of the first n steps of the last run (option "trace_steps" = n):
        flags bit 0 accepted, bit 1 overlap, bit 2 rotation."""
        d = np.zeros((self.R, int(n_steps)))
        f = np.zeros((self.R, int(n_steps)), dtype=np.uint8)
        check(self._L.mmc_batch_get_trace(self._h, _d(d), f.ctypes.data_as(C.POINTER(C.c_uint8))))
        return d, f

    def set_orientations(self, quat, db, faithful=True):
        """Turn on the reference's quaternion move generation for device-side proposals
        (mmc_batch_set_orientations): quat (n_mol, 4), db (3, 3) body-fixed sites; faithful keeps
        the reference's q_to_a including its (2,3) element.  quat=None turns it off."""
        if quat is None:
            check(self._L.mmc_batch_set_orientations(self._h, None, None, 0))
            return
        q, d = _f64(quat).reshape(self.n_mol, 4), _f64(db).reshape(3, 3)
        check(self._L.mmc_batch_set_orientations(self._h, _d(q), _d(d), 1 if faithful else 2))

    def get_orientations(self, r):
        q = np.empty((self.n_mol, 4))
        check(self._L.mmc_batch_get_orientations(self._h, int(r), _d(q)))
        return q

    def peek_part(self, r, part):
        """(64 raw bytes, stamp) of the result record of (replica r, part) of the last eval()."""
        buf = (C.c_uint8 * 64)()
        stamp = C.c_uint32()
        check(self._L.mmc_batch_peek_part(self._h, int(r), int(part), buf, C.byref(stamp)))
        return bytes(buf), stamp.value

    def part_validate(self, raw64, stamp):
        buf = (C.c_uint8 * 64).from_buffer_copy(raw64)
        return bool(self._L.mmc_part_validate(buf, int(stamp)))

    def rdf(self, site, numbins):
        """mmc_batch_rdf: histogram hist[0..numbins] of gr.jl's makeRDF over all replicas."""
        hist = np.zeros(int(numbins) + 1, dtype=np.uint64)
        check(self._L.mmc_batch_rdf(self._h, int(site), int(numbins),
                                    hist.ctypes.data_as(C.POINTER(C.c_uint64))))
        return hist

    def new_chains(self, energies, virials=None, dr_max=0.15, dphi_max=0.05, set_value=0.5):
        """One mmc_chain record per replica (numpy structured array, _lib.CHAIN_DTYPE): the
        bookkeeping Loop() keeps in total / averages / trans_moves / rot_moves / totProps."""
        c = np.zeros(self.R, dtype=CHAIN_DTYPE)
        c["dr_max"], c["dphi_max"] = dr_max, dphi_max
        c["energy"] = energies
        c["virial"] = 0.0 if virials is None else virials
        c["trans_set_value"] = c["rot_set_value"] = set_value
        return c

    def run_chains(self, chains, n_steps, temperature, seed, adjust=True, n_groups=2, n_parts=0,
                   time_kernels=False, n_threads=1, n_streams=0, replica0=0):
        """mmc_batch_run_chains: `chains` (from new_chains) is updated in place."""
        if chains.dtype != CHAIN_DTYPE or chains.shape != (self.R,) or not chains.flags.c_contiguous:
            raise ValueError("chains must be the array returned by new_chains()")
        p = RunParams(float(temperature), 0.0, 0.0, int(seed), int(n_steps), int(n_groups),
                      int(n_parts), int(time_kernels), int(n_threads), int(n_streams), 0,
                      int(replica0))
        st = RunStats()
        check(self._L.mmc_batch_run_chains(self._h, C.byref(p), chains.ctypes.data_as(C.c_void_p),
                                           int(bool(adjust)), C.byref(st)))
        return st.asdict()
