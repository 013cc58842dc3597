"""CPU test of the N>1 path: two processes over gloo (the GPU box runs the same code over RCCL).

Each rank runs the chains of its shard -- here driven by the CPU oracle, since there is no GPU --
seeded by GLOBAL replica index, then the observables are all-reduced.  The reduced result must
equal a single-process run over all chains: sharding changes neither the chains nor the sums."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_chains(indices, n_moves=12):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    from metropolismontecarlo_amd import sharding
    from oracle import oracle as orc
    a = common.nist_arrays(1, "unwrapped")
    out = dict(moves=0, accepted=0, overlaps=0, energy_sum=0.0, kernel_ms=0.0, launches=0)
    for g in indices:
        s = common.oracle_system(a)
        ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
        e = orc.potential_ewald(s, ew, 10.0, 10.0)["energy"]
        rng = np.random.default_rng(sharding.replica_seed(g))
        for n in range(n_moves):
            i = n % s.n_mol + 1
            d = (rng.random(3) - 0.5) * 0.316555789
            cn, an = s.com[i - 1] + d, s.coords[3 * (i - 1):3 * i] + d
            dd, ov = orc.trial_move(i, s, ew, 10.0, 10.0, cn, an)
            delta = dd[0] + dd[1] + dd[2]
            out["moves"] += 1
            out["overlaps"] += int(ov)
            if not ov and (delta < 0 or np.exp(-delta / 298.15) > rng.random()):
                e += delta
                s.com[i - 1], s.coords[3 * (i - 1):3 * i] = cn, an
                ew.sumQExpOld = ew.sumQExpNew.copy()
                out["accepted"] += 1
            else:
                ew.sumQExpNew = ew.sumQExpOld.copy()
        out["energy_sum"] += e
        out["launches"] += n_moves
    return out


def _worker(rank, world, port, per_gpu, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from metropolismontecarlo_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, lr, w = sharding.env_rank()
    assert (r, w) == (rank, world)
    local = run_chains(sharding.shard(per_gpu, r))
    red, tmax = sharding.reduce_observables(local, elapsed=1.0 + r, dist=dist)
    dist.barrier()
    if rank == 0:
        q.put((red, tmax))
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    sys.path.insert(0, ROOT)
    from metropolismontecarlo_amd import sharding
    world, per_gpu = 2, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    red, tmax = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = run_chains(range(world * per_gpu))
    for k in sharding.OBSERVABLES:
        assert red[k] == pytest.approx(single[k], rel=1e-14, abs=1e-12), k
    assert tmax == 2.0                                   # MAX over ranks of 1.0 + rank
    assert red["moves"] == world * per_gpu * 12 and 0 < red["accepted"] <= red["moves"]


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    from metropolismontecarlo_amd import sharding
    assert list(sharding.shard(4, 2)) == [8, 9, 10, 11]
    # BASELINE configs[2]: 256 replicas over 8 GPUs -> 32 each, disjoint and complete
    cover = [g for r in range(8) for g in sharding.shard_total(256, r, 8)]
    assert cover == list(range(256))
    assert [len(sharding.shard_total(10, r, 4)) for r in range(4)] == [3, 3, 2, 2]
    assert sharding.replica_seed(0) == 11234 and sharding.replica_seed(5, phase=1) != sharding.replica_seed(5)
    red, t = sharding.reduce_observables(dict(moves=3, accepted=1, overlaps=0, energy_sum=-1.5,
                                              kernel_ms=0.2, launches=3), 0.5)
    assert red["energy_sum"] == -1.5 and t == 0.5
