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


# ---- host threads of a rank and where they run -------------------------------------------------
# Every rank spins host threads (the accept/reject workers of mmc_batch_run; on the move-server
# path they write control words through the PCIe BAR).  On an 8-GPU node eight ranks x up to 8
# spinning threads must neither oversubscribe the cores nor sit on the other socket from their GPU.
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of sysfs cpulist files)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(out))


def gpu_numa_node(domain, bus, device, sysfs="/sys"):
    """NUMA node of the GPU at PCI address domain:bus:device.0, or -1 when the system does not say."""
    path = os.path.join(sysfs, "bus", "pci", "devices", "%04x:%02x:%02x.0" % (domain, bus, device),
                        "numa_node")
    try:
        with open(path) as fh:
            return int(fh.read().strip())
    except (OSError, ValueError):
        return -1


def node_cpus(node, sysfs="/sys"):
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as fh:
            return parse_cpulist(fh.read())
    except OSError:
        return []


def plan_host_threads(local_rank, gpu_nodes, cpus_of_node, allowed, cap=8):
    """Which cores rank `local_rank` pins itself to and how many worker threads it runs.

    gpu_nodes[k]: NUMA node of local GPU k (-1 unknown), one entry per local rank; cpus_of_node:
    {node: [cpu, ...]}; allowed: the cores this process may use.  The ranks whose GPUs sit on one
    node split that node's allowed cores into disjoint contiguous slices (ranks of unknown node
    split all allowed cores among all ranks); a rank runs min(cap, slice - 1) threads, at least 1
    -- one core of the slice stays free for the launching thread of the next rank's neighbour,
    the OS and RCCL's proxy thread.  Returns (sorted cpus, n_threads)."""
    allowed = sorted(allowed)
    node = gpu_nodes[local_rank]
    pool = [c for c in cpus_of_node.get(node, []) if c in set(allowed)] if node >= 0 else []
    if pool:
        peers = [k for k, n in enumerate(gpu_nodes) if n == node]
    else:                       # unknown placement: all ranks share everything evenly
        pool, peers = allowed, list(range(len(gpu_nodes)))
    k, n = peers.index(local_rank), len(peers)
    lo, hi = (len(pool) * k) // n, (len(pool) * (k + 1)) // n
    mine = pool[lo:hi] or pool[:1] or allowed[:1]
    return mine, max(1, min(cap, len(mine) - 1))


def pin_rank_to_gpu_numa(local_rank, pci_addresses, cap=8, sysfs="/sys"):
    """os.sched_setaffinity of THIS process (threads created later inherit it) to its slice of the
    cores next to its GPU.  pci_addresses[k] = (domain, bus, device) of local GPU k.  No exec, no
    re-launch.  Returns (cpus, n_threads)."""
    allowed = sorted(os.sched_getaffinity(0))
    nodes = [gpu_numa_node(*a, sysfs=sysfs) for a in pci_addresses]
    cpus = {n: node_cpus(n, sysfs) for n in set(nodes) if n >= 0}
    mine, n_threads = plan_host_threads(local_rank, nodes, cpus, allowed, cap)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        mine = allowed
    return mine, n_threads


def default_id_path():
    """Where rank 0 leaves the RCCL unique id for the other ranks of ONE run on one node:
    MMC_DIST_ID_FILE if set, else a name keyed on what the ranks of a launch share and two launches
    do not -- the user, MASTER_PORT, the launcher's run id and the launcher's process id (every
    rank of a `torch.distributed.run` / spawn launch has the same parent)."""
    explicit = os.environ.get("MMC_DIST_ID_FILE")
    if explicit:
        return explicit
    token = "%d.%s.%s.%d" % (os.getuid(), os.environ.get("MASTER_PORT", "0"),
                             os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "mmc_dist_id." + token)


ID_MAX_AGE_S = 120.0   # a file older than this was left by a run that died: never this run's id


def id_file_exchange(rank, raw, path=None, timeout_s=60.0):
    """Rank 0 (raw = its 128 id bytes) publishes, every other rank (raw = None) waits for and
    returns them.  Rank 0 removes whatever is at `path` first and writes atomically (rename);
    readers ignore a file older than ID_MAX_AGE_S, so a leftover of a dead run is not taken for
    this run's id.  Call id_file_done(path) on rank 0 once every rank has joined the communicator
    (mmc_dist_init returns then)."""
    import time
    path = path or default_id_path()
    if raw is not None:
        if rank != 0:
            raise ValueError("only rank 0 publishes the id")
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as fh:
            fh.write(raw)
        os.replace(tmp, path)
        return raw
    deadline = time.monotonic() + timeout_s
    while time.monotonic() < deadline:
        try:
            st = os.stat(path)
            if st.st_size == 128 and time.time() - st.st_mtime < ID_MAX_AGE_S:
                with open(path, "rb") as fh:
                    data = fh.read()
                if len(data) == 128:
                    return data
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise RuntimeError(f"no unique id from rank 0 at {path} within {timeout_s:.0f} s")


def id_file_done(path=None):
    try:
        os.unlink(path or default_id_path())
    except FileNotFoundError:
        pass


class RcclReducer:
    """The reduction over the library's own RCCL communicator (mmc_dist_*, include/mmc_hip.h): what
    a host without torch uses.  Rank 0 creates the unique id; `exchange(id_bytes_or_None)` must
    hand every rank rank 0's 128 bytes (default: id_file_exchange, a per-run file on the node)."""

    def __init__(self, rank, world, device=0, exchange=None):
        import ctypes as C
        from . import _lib
        self._L, self._C = _lib.lib(), C
        ident = C.create_string_buffer(128)
        own_file = exchange is None and world > 1
        if exchange is None:
            def exchange(raw):
                return id_file_exchange(rank, raw)
        if rank == 0:
            _lib.check(self._L.mmc_dist_unique_id(ident))
        raw = exchange(ident.raw if rank == 0 else None) if world > 1 else ident.raw
        self._h = C.c_void_p()
        try:
            _lib.check(self._L.mmc_dist_init(rank, world, C.create_string_buffer(raw, 128), device,
                                             C.byref(self._h)))
        finally:
            if own_file and rank == 0:   # every rank has read it (init is collective) or the run is lost
                id_file_done()
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
