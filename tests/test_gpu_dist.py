"""The product's N>1 launch path, rehearsed on one GPU: the driver's own command line
`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` with
MMC_DIST_BACKEND=gloo, so that both ranks share cuda:0 and the final reduction runs over gloo (on
the 8-GPU node the same code runs one rank per GPU over RCCL).  Launched by tests/conftest.py
before this process touches the GPU; checked here:

  * the JSON contract line of rank 0 (n_gpus, moves, weak scaling, roofline object),
  * running totals equal a recompute on every rank (energy_drift_rel),
  * the 2 x 64 chains are THE SAME chains as one process running global indices 0..127:
    trajectories depend on (seed, global replica index) only -- identical accept counts and the
    same energy sum.

No scaling claim is made from this: two ranks share one device."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def load(out, name):
    rc = open(os.path.join(out, name + ".rc")).read().strip()
    err = open(os.path.join(out, name + ".err")).read()
    assert rc == "0", (name, rc, err[-2000:])
    lines = [l for l in open(os.path.join(out, name + ".json")).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, (name, lines, err[-2000:])      # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_two_ranks_share_nothing_but_the_final_reduction(dist_rehearsal):
    two, one = load(dist_rehearsal, "two"), load(dist_rehearsal, "one")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["scaling"] == "weak" and two["unit"] == "moves/s" and two["dtype"] == "f64"
    assert two["steps"] == 20 and two["warmup"] == 5
    assert two["config"]["replicas_per_gpu"] == 64 and two["config"]["replicas_total"] == 128
    assert one["config"]["replicas_total"] == 128
    moves = 2 * 64 * 20
    assert abs(two["value"] * two["ms_per_step"] * 1e-3 * 20 - moves) < 1e-6 * moves   # value = moves / time
    for d in (two, one):
        assert d["energy_drift_rel"] < 1e-12
        assert d["roofline"]["kernel"] in ("k_move_eval_wave", "k_move_eval_fast", "k_move_eval_lat")
        assert d["roofline"]["frac"] > 0
        assert d["vs_baseline"] is None and d["higher_is_better"] is True
    # the same 128 chains either way
    assert two["acceptance"] == one["acceptance"]
    assert two["overlaps"] == one["overlaps"]
    e2, e1 = two["energy_mean_per_replica"], one["energy_mean_per_replica"]
    assert abs(e2 - e1) < 1e-13 * abs(e1), (e2, e1)


def test_two_ranks_of_32_chains_run_the_move_server(dist_rehearsal):
    """BASELINE configs[2] is 256 chains over 8 GPUs = 32 per rank: that shape runs on the
    persistent move server (host threads answering control words), here with two ranks at once."""
    d = load(dist_rehearsal, "two32")
    assert d["n_gpus"] == 2 and d["config"]["replicas_per_gpu"] == 32 and d["config"]["replicas_total"] == 64
    assert "move server" in d["config"]["driver"]
    assert d["energy_drift_rel"] < 1e-12 and d["torn_result_records"] == 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * 200 - 64 * 200) < 1e-6 * 64 * 200
    assert 0.5 < d["acceptance"] < 0.95


def test_two_ranks_in_the_headline_s_mode_run_the_same_chains_as_one(dist_rehearsal):
    """The mode the headline runs in -- the move kernel takes the accept decision, a launch takes a
    group through eight steps -- under two ranks of 10240 chains, against one process running global
    indices 0..20479: the same chains (accept count, energy sum), whatever the split."""
    two, one = load(dist_rehearsal, "two_dev"), load(dist_rehearsal, "one_dev")
    for d in (two, one):
        assert d["config"]["accept_decision"] == "move kernel" and d["roofline"]["steps_per_launch"] == 8
        assert d["roofline"]["kernel"] == "k_move_eval_wave"
        assert d["energy_drift_rel"] < 1e-12 and d["torn_result_records"] == 0
        assert d["config"]["replicas_total"] == 20480 and d["steps"] == 24
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["acceptance"] == one["acceptance"] and two["overlaps"] == one["overlaps"]
    e2, e1 = two["energy_mean_per_replica"], one["energy_mean_per_replica"]
    assert abs(e2 - e1) < 1e-13 * abs(e1), (e2, e1)


def test_rccl_collective_behind_the_c_abi():
    """mmc_dist_*: the final reduction for hosts without torch.  A one-GPU box can form a
    communicator of ONE rank only (RCCL wants a GPU per rank): that still goes through
    ncclGetUniqueId / ncclCommInitRank / ncclAllReduce (sum and max, fp64) / ncclCommDestroy."""
    import ctypes as C
    import numpy as np
    from metropolismontecarlo_amd import _lib
    L = _lib.lib()
    ident = C.create_string_buffer(128)
    _lib.check(L.mmc_dist_unique_id(ident))
    assert any(ident.raw)
    d = C.c_void_p()
    _lib.check(L.mmc_dist_init(0, 1, ident, 0, C.byref(d)))
    sums = np.array([1.5, -2.25, 3e10, 0.0])
    mx = np.array([0.125, -7.0])
    dp = C.POINTER(C.c_double)
    for _ in range(3):
        _lib.check(L.mmc_dist_reduce(d, sums.ctypes.data_as(dp), 4, mx.ctypes.data_as(dp), 2))
    assert np.array_equal(sums, [1.5, -2.25, 3e10, 0.0]) and np.array_equal(mx, [0.125, -7.0])
    _lib.check(L.mmc_dist_reduce(d, None, 0, mx.ctypes.data_as(dp), 2))
    _lib.check(L.mmc_dist_destroy(d))
    # the same through sharding.reduce_observables
    from metropolismontecarlo_amd import sharding
    local = dict(moves=10, accepted=4, overlaps=0, energy_sum=-1.5, kernel_ms=2.0, launches=3)
    with sharding.RcclReducer(0, 1) as red:
        out, t = sharding.reduce_observables(local, 0.25, red)
    assert out["moves"] == 10 and out["energy_sum"] == -1.5 and t == 0.25
