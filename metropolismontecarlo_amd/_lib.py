"""ctypes binding of libmmc_hip.so -- every symbol declared in include/mmc_hip.h.

There is no CPU fallback anywhere in this package: if the shared library is missing the import
fails loudly, and if no HIP device is visible every compute entry point raises MMCError with the
library's message.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMC_HIP_LIB: another build of the same library (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get("MMC_HIP_LIB") or os.path.join(_HERE, "libmmc_hip.so")

MMC_OK, MMC_ERR_ARG, MMC_ERR_ASSERT, MMC_ERR_HIP, MMC_ERR_STATE, MMC_ERR_UNSUPPORTED = range(6)
_STATUS_NAME = {1: "MMC_ERR_ARG", 2: "MMC_ERR_ASSERT", 3: "MMC_ERR_HIP", 4: "MMC_ERR_STATE",
                5: "MMC_ERR_UNSUPPORTED"}


class MMCError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{_STATUS_NAME.get(status, status)}: {message}")
        self.status = status


class Totals(C.Structure):
    """mmc_totals: the Properties fields potential() fills + the four terms it prints."""
    _fields_ = [("energy", C.c_double), ("virial", C.c_double), ("coulomb", C.c_double),
                ("lj", C.c_double), ("real", C.c_double), ("recip", C.c_double),
                ("self", C.c_double), ("n_overlap", C.c_int32), ("_pad", C.c_int32)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "_pad"}


class Move(C.Structure):
    """mmc_move"""
    _fields_ = [("mol", C.c_int32), ("accept_prev", C.c_int32), ("com_new", C.c_double * 3),
                ("atoms_new", C.c_double * 9)]


class MoveResult(C.Structure):
    """mmc_move_result"""
    _fields_ = [("d_lj", C.c_double), ("d_real", C.c_double), ("d_recip", C.c_double),
                ("d_vir", C.c_double), ("overlap", C.c_int32), ("_pad", C.c_int32)]


class RunParams(C.Structure):
    """mmc_run_params"""
    _fields_ = [("temperature", C.c_double), ("dr_max", C.c_double), ("dphi_max", C.c_double),
                ("seed", C.c_uint64), ("n_steps", C.c_int64), ("n_groups", C.c_int32),
                ("n_parts", C.c_int32), ("time_kernels", C.c_int32), ("n_threads", C.c_int32),
                ("n_streams", C.c_int32), ("_pad", C.c_int32), ("replica0", C.c_uint64)]


class RunStats(C.Structure):
    """mmc_run_stats"""
    _fields_ = [("moves", C.c_int64), ("launches", C.c_int64), ("trans_attempt", C.c_int64),
                ("trans_accept", C.c_int64), ("rot_attempt", C.c_int64),
                ("rot_accept", C.c_int64), ("overlaps", C.c_int64), ("wall_ms", C.c_double),
                ("kernel_ms", C.c_double), ("energy_sum", C.c_double),
                ("timed_launches", C.c_int64), ("torn_records", C.c_int64),
                ("server_steps", C.c_int64), ("device_decisions", C.c_int64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class NptParams(C.Structure):
    """mmc_npt_params"""
    _fields_ = [("pressure", C.c_double), ("vmax", C.c_double), ("alpha", C.c_double),
                ("n_sweeps", C.c_int64), ("moves_per_sweep", C.c_int64)]


class NptStats(C.Structure):
    """mmc_npt_stats"""
    _fields_ = [("vol_attempt", C.c_int64), ("vol_accept", C.c_int64), ("box", C.c_double),
                ("volume_sum", C.c_double), ("volume_ms", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# mmc_totals as a numpy record (Batch.potential_ewald(as_array=True): no per-replica Python objects)
TOTALS_DTYPE = np.dtype([("energy", "f8"), ("virial", "f8"), ("coulomb", "f8"), ("lj", "f8"),
                         ("real", "f8"), ("recip", "f8"), ("self", "f8"), ("n_overlap", "i4"),
                         ("_pad", "i4")])

# mmc_chain as a numpy record: one row per replica, passed by pointer
CHAIN_DTYPE = np.dtype([("dr_max", "f8"), ("dphi_max", "f8"), ("energy", "f8"), ("virial", "f8"),
                        ("avg_energy", "f8"), ("avg_virial", "f8"), ("steps_taken", "i8"),
                        ("overlaps", "i8"), ("trans_naccepp", "i8"), ("trans_attempp", "i8"),
                        ("trans_naccept", "i8"), ("trans_attempt", "i8"), ("rot_naccepp", "i8"),
                        ("rot_attempp", "i8"), ("rot_naccept", "i8"), ("rot_attempt", "i8"),
                        ("trans_set_value", "f8"), ("rot_set_value", "f8")])

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p
_d = C.c_double
_i64 = C.c_int64
_i32 = C.c_int32

# name -> argtypes (restype is int32 status unless listed in _RESTYPE)
SIGNATURES = {
    "mmc_device_count": [_i32p],
    "mmc_ctx_create": [_i32, _vp, C.POINTER(_vp)],
    "mmc_ctx_destroy": [_vp],
    "mmc_ctx_synchronize": [_vp],
    "mmc_upload_system": [_vp, _i64, _i64, _dp, _i64p, _i64p, _dp, _i64p, _dp, _i64, _dp, _dp, _d],
    "mmc_set_molecule": [_vp, _i64, _dp, _dp],
    "mmc_update_system": [_vp, _dp, _dp],
    "mmc_download_system": [_vp, _dp, _dp],
    "mmc_volume_change": [_vp, _d, _d],
    "mmc_batch_volume_change": [_vp, _d, _d],
    "mmc_batch_volume_trial": [_vp, _d, _d, C.POINTER(Totals)],
    "mmc_batch_volume_accept": [_vp],
    "mmc_batch_volume_reject": [_vp],
    "mmc_batch_run_npt": [_vp, C.POINTER(RunParams), C.POINTER(NptParams), _dp, C.POINTER(RunStats),
                          C.POINTER(NptStats)],
    "mmc_volume_trial": [_vp, _d, _d, _d, _d, C.POINTER(Totals)],
    "mmc_volume_accept": [_vp],
    "mmc_volume_reject": [_vp],
    "mmc_prepare_ewald": [_vp, _d, _i64, _i64, _d, _d, _i64p],
    "mmc_get_kvectors": [_vp, _i32p, _dp],
    "mmc_get_sumqexp": [_vp, _dp, _dp],
    "mmc_set_sumqexp": [_vp, _dp, _dp],
    "mmc_lj_poly_du": [_vp, _i64, _d, _dp, _dp],
    "mmc_ewald_real": [_vp, _i64, _d, _d, _dp, _i32p],
    "mmc_ewald_short": [_vp, _i64, _d, _dp, _dp, _i32p],
    "mmc_coulomb_real": [_vp, _i64, _d, _dp, _i32p],
    "mmc_recip_long": [_vp, _dp],
    "mmc_recip_move": [_vp, _dp, _dp, _dp, _i64, _dp],
    "mmc_call_lj_poly_du": [_vp, _i64, _vp, _vp, _d, _dp, _dp],
    "mmc_call_ewald_real": [_vp, _i64, _vp, _vp, _d, _d, _dp, _i32p],
    "mmc_call_ewald_short": [_vp, _i64, _vp, _vp, _d, _dp, _dp, _i32p],
    "mmc_call_recip_move": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _dp],
    "mmc_ctx_stats": [_vp, _i64p],
    "mmc_ctx_ping": [_vp, _i64, _dp],
    "mmc_ctx_set_option": [_vp, C.c_char_p, _i64],
    "mmc_recip_commit": [_vp],
    "mmc_recip_rollback": [_vp],
    "mmc_ewald_self": [_vp, _dp],
    "mmc_potential_ewald": [_vp, _d, _d, C.POINTER(Totals)],
    "mmc_potential_wolf": [_vp, _d, _d, C.POINTER(Totals)],
    "mmc_trial_move": [_vp, _i64, _dp, _dp, _d, _d, _dp, _i32p],
    "mmc_accept_move": [_vp],
    "mmc_reject_move": [_vp],
    "mmc_batch_create": [_i32, _vp, _i64, _i64, _dp, _dp, _i64p, _dp, _i64, _dp, _dp, _d, _d,
                         _i64, _i64, _d, _d, _d, C.POINTER(_vp)],
    "mmc_batch_destroy": [_vp],
    "mmc_batch_set_replica": [_vp, _i64, _dp, _dp],
    "mmc_batch_get_replica": [_vp, _i64, _dp, _dp, _dp],
    "mmc_batch_recip_long": [_vp, _dp],
    "mmc_batch_potential_ewald": [_vp, _vp],  # mmc_totals[R]: a ctypes array or a numpy buffer
    "mmc_batch_eval": [_vp, C.POINTER(Move), C.POINTER(MoveResult)],
    "mmc_batch_set_parts": [_vp, _i32],
    "mmc_batch_set_option": [_vp, C.c_char_p, _i64],
    "mmc_batch_qq_table": [_vp, _dp, _i64, _dp],
    "mmc_batch_settle": [_vp, _i32p],
    "mmc_batch_get_trace": [_vp, _dp, C.POINTER(C.c_uint8)],
    "mmc_batch_set_orientations": [_vp, _dp, _dp, _i32],
    "mmc_batch_get_orientations": [_vp, _i64, _dp],
    "mmc_part_validate": [_vp, C.c_uint32],
    "mmc_batch_peek_part": [_vp, _i64, _i32, _vp, C.POINTER(C.c_uint32)],
    "mmc_batch_rdf": [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)],
    "mmc_batch_run": [_vp, C.POINTER(RunParams), _dp, C.POINTER(RunStats)],
    "mmc_study_f32_total": [_vp, C.c_double, C.c_double, C.c_int32, _dp],
    "mmc_study_f32_move": [_vp, _i64, _dp, _dp, C.c_double, C.c_double, C.c_int32, _dp, _i32p],
    "mmc_philox4x32": [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
    "mmc_batch_run_chains": [_vp, C.POINTER(RunParams), _vp, C.c_int32, C.POINTER(RunStats)],
    "mmc_chain_block_line": [_vp, _i64, _i64, _d, _d, C.c_char_p, _i64],
    "mmc_dist_unique_id": [C.c_char_p],                      # uint8_t[128]
    "mmc_dist_init": [_i32, _i32, C.c_char_p, _i32, C.POINTER(_vp)],
    "mmc_dist_reduce": [_vp, _dp, _i64, _dp, _i64],
    "mmc_dist_destroy": [_vp],
}
_RESTYPE = {"mmc_last_error": C.c_char_p, "mmc_version": C.c_char_p}

_lib = None


def lib():
    """The loaded library.  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m metropolismontecarlo_amd.build` "
                "(hipcc, gfx950).  This package has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == header and library disagree
            fn.argtypes = args
            fn.restype = C.c_int32
        for name, rt in _RESTYPE.items():
            fn = getattr(L, name)
            fn.argtypes = []
            fn.restype = rt
        _lib = L
    return _lib


def check(status):
    if status != MMC_OK:
        msg = lib().mmc_last_error()
        msg = msg.decode() if msg else ""
        if status == MMC_ERR_ASSERT:
            raise AssertionError(msg)  # the reference's @assert throws AssertionError
        raise MMCError(status, msg)


def exported_symbols():
    return sorted(list(SIGNATURES) + list(_RESTYPE))
