"""Multi-GPU layout of the replica ensemble: one process per GPU, replicas sharded by global
index, no data-path collective.  The only collective (C1 in SURVEY.md) is the final reduction of a
handful of observables -- sums of energies / acceptance counters and the max of the elapsed time --
over RCCL (backend "nccl" on the GPU box, "gloo" in the CPU tests)."""
import os

import numpy as np

BASE_SEED = 11234  # Monatomic/mainMonatomic.jl:15, the only seed the reference fixes

OBSERVABLES = ("moves", "accepted", "overlaps", "energy_sum", "kernel_ms", "launches")


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard(replicas_per_gpu, rank):
    """Weak scaling: every rank owns `replicas_per_gpu` chains; global index = rank*R + r."""
    lo = rank * replicas_per_gpu
    return range(lo, lo + replicas_per_gpu)


def shard_total(total_replicas, rank, world):
    """Strong scaling (BASELINE configs[2]: 256 replicas over 8 GPUs): contiguous blocks, the
    first `total % world` ranks take one extra."""
    base, extra = divmod(total_replicas, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def replica_seed(global_index, phase=0):
    """Seed of a host-side (numpy) stream of one chain: depends on its GLOBAL index only, so a
    chain's trajectory does not depend on how many GPUs the ensemble is spread over.  `phase`
    separates warm-up from the timed run.  (The native driver does not add indices to seeds: it
    keys its streams by the pair (seed, global replica index) -- pass run_seed(phase) as `seed`
    and the shard's first global index as `replica0`.)"""
    return BASE_SEED + int(global_index) + 1_000_003 * int(phase)


def run_seed(phase=0):
    """`seed` of mmc_batch_run for one phase of a run; the same on every rank."""
    return BASE_SEED + 1_000_003 * int(phase)


class RcclReducer:
    """The reduction over the library's own RCCL communicator (mmc_dist_*, include/mmc_hip.h): what
    a host without torch uses.  Rank 0 creates the unique id; `exchange(id_bytes_or_None)` must
    hand every rank rank 0's 128 bytes (default: a file named by MMC_DIST_ID_FILE)."""

    def __init__(self, rank, world, device=0, exchange=None):
        import ctypes as C
        import time
        from . import _lib
        self._L, self._C = _lib.lib(), C
        ident = C.create_string_buffer(128)
        if exchange is None:
            path = os.environ.get("MMC_DIST_ID_FILE", "/tmp/mmc_dist_id")

            def exchange(raw):
                if raw is not None:
                    with open(path + ".tmp", "wb") as fh:
                        fh.write(raw)
                    os.replace(path + ".tmp", path)
                    return raw
                for _ in range(6000):
                    if os.path.exists(path) and os.path.getsize(path) == 128:
                        return open(path, "rb").read()
                    time.sleep(0.01)
                raise RuntimeError("no unique id from rank 0")
        if rank == 0:
            _lib.check(self._L.mmc_dist_unique_id(ident))
        raw = exchange(ident.raw if rank == 0 else None) if world > 1 else ident.raw
        self._h = C.c_void_p()
        _lib.check(self._L.mmc_dist_init(rank, world, C.create_string_buffer(raw, 128), device,
                                         C.byref(self._h)))
        self.world = world

    def reduce(self, sums, maxima):
        from . import _lib
        dp = self._C.POINTER(self._C.c_double)
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        maxima = np.ascontiguousarray(maxima, dtype=np.float64)
        _lib.check(self._L.mmc_dist_reduce(self._h, sums.ctypes.data_as(dp), len(sums),
                                           maxima.ctypes.data_as(dp), len(maxima)))
        return sums, maxima

    def close(self):
        if self._h:
            self._L.mmc_dist_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def reduce_observables(local, elapsed, dist=None, device="cpu"):
    """SUM the observable vector and MAX the elapsed time over all ranks.  `dist` is
    torch.distributed (initialised), an RcclReducer, or None for a single process."""
    vec = np.array([float(local[k]) for k in OBSERVABLES], dtype=np.float64)
    if isinstance(dist, RcclReducer):
        s, m = dist.reduce(vec, [float(elapsed)])
        return dict(zip(OBSERVABLES, s)), float(m[0])
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(zip(OBSERVABLES, vec)), float(elapsed)
    import torch
    t = torch.tensor(vec, dtype=torch.float64, device=device)
    m = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return dict(zip(OBSERVABLES, t.cpu().numpy())), float(m.item())
