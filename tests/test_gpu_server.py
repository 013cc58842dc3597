"""The persistent move server (k_move_server_wave, option "persistent") against the launch-per-step
driver (SURVEY.md section 8 rows a/b: Loop()'s move body, Ewald/main.jl:487-644).

The server evaluates a move with the same part split and the same summation order as a launch with
n_parts = its waves, draws the same counter-based proposals and leaves accept/reject to the same host
code, so everything the two drivers return must be IDENTICAL bit for bit: energies, statistics,
chain records, coordinates and structure factors.  The launch-per-step driver itself is pinned to
the oracle by test_gpu_batch.py / test_gpu_moves.py."""
import numpy as np
import pytest

import common
from metropolismontecarlo_amd import structs
from metropolismontecarlo_amd._lib import MMCError
from metropolismontecarlo_amd.device import Batch

pytestmark = pytest.mark.gpu


def make_batch(a, R, persistent, kernel=2):
    b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
              5.6 / a["box"], structs.factor, 10.0, 10.0)
    b.set_option("device_moves", 1)
    b.set_option("kernel", kernel)
    b.set_option("persistent", persistent)
    return b


def server_waves(n_mol):
    return min((n_mol + 63) // 64, 7) + 1


def state(b):
    out = []
    for r in range(b.R):
        out.extend(b.get_replica(r))
    return out


def same(xs, ys):
    return len(xs) == len(ys) and all(np.array_equal(x, y) for x, y in zip(xs, ys))


@pytest.mark.parametrize("k,R,steps,threads", [(1, 1, 230, 1), (1, 3, 205, 2), (4, 1, 400, 1),
                                               (4, 32, 160, 2), (2, 5, 60, 3)])
def test_server_equals_launch_per_step(k, R, steps, threads):
    """mmc_batch_run twice in a row (the second call continues the random streams and the
    S-buffer parity of the first), fixed step sizes."""
    a = common.nist_arrays(k, "unwrapped")
    P = server_waves(a["com"].shape[0])
    res = []
    for persistent in (1, 0):
        with make_batch(a, R, persistent) as b:
            e = b.potential_ewald(as_array=True)["energy"].copy()
            stats = []
            for n in (steps, 37):
                e, st = b.run(n, 298.15, 0.316555789, 0.05, 11, e, n_groups=min(R, 2), n_parts=P,
                              n_threads=threads)
                stats.append([st[q] for q in ("moves", "trans_accept", "rot_accept", "overlaps",
                                              "trans_attempt", "rot_attempt")])
            assert st["torn_records"] == 0
            assert st["launches"] == 37 * min(R, 2)
            assert st["server_steps"] == (37 if persistent else 0)
            # the accumulated energy still is the energy of the configuration
            e2 = b.potential_ewald(as_array=True)["energy"]
            assert np.abs(e - e2).max() <= 1e-11 * np.abs(e2).max()
            res.append((e, stats, state(b)))
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1]
    assert res[0][1][0][1] + res[0][1][0][2] > 0          # something was accepted
    assert same(res[0][2], res[1][2])


@pytest.mark.parametrize("k,R,parts,steps", [(4, 1, 16, 400), (4, 2, 8, 150), (1, 1, 8, 230),
                                             (2, 3, 12, 90), (4, 1, 12, 300)])
def test_latency_server_equals_launch_per_step(k, R, parts, steps):
    """The latency server (k_move_server_lat: parts / 4 workgroups per replica, each reading the
    replica's control word itself and keeping its own copy of its molecules) against
    k_move_eval_lat launched per step with the same part count: identical chains bit for bit,
    over two calls (random streams and S-buffer parity continue)."""
    a = common.nist_arrays(k, "unwrapped")
    res = []
    for persistent in (1, 0):
        with make_batch(a, R, persistent, kernel=4) as b:
            e = b.potential_ewald(as_array=True)["energy"].copy()
            stats = []
            for n in (steps, 41):
                e, st = b.run(n, 298.15, 0.316555789, 0.05, 17, e, n_groups=1, n_parts=parts)
                stats.append([st[q] for q in ("moves", "trans_accept", "rot_accept", "overlaps")])
            assert st["torn_records"] == 0 and st["server_steps"] == (41 if persistent else 0)
            e2 = b.potential_ewald(as_array=True)["energy"]
            assert np.abs(e - e2).max() <= 1e-11 * np.abs(e2).max()
            res.append((e, stats, state(b)))
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    assert res[0][1][0][1] + res[0][1][0][2] > 0
    assert same(res[0][2], res[1][2])


def test_single_chain_takes_the_latency_server_by_default():
    """One chain (BASELINE configs[1]): the automatic choice is the latency server with four
    workgroups; it samples the same chain as an explicit request for it."""
    a = common.nist_arrays(4, "unwrapped")
    res = []
    for kernel, parts in ((3, 0), (4, 16)):
        with make_batch(a, 1, -1, kernel=kernel) as b:
            e = b.potential_ewald(as_array=True)["energy"].copy()
            e, st = b.run(300, 298.15, 0.316555789, 0.05, 5, e, n_groups=1, n_parts=parts)
            assert st["server_steps"] == 300
            e2 = b.potential_ewald(as_array=True)["energy"]
            assert np.abs(e - e2).max() <= 1e-11 * np.abs(e2).max()
            res.append((e, state(b)))
    assert np.array_equal(res[0][0], res[1][0]) and same(res[0][1], res[1][1])


def test_server_chains_with_step_adjustment():
    """mmc_batch_run_chains with Adjust! every sweep (adjust.jl:1-83): the server re-reads the step
    sizes when told to and re-draws its speculative proposal."""
    a = common.nist_arrays(1, "unwrapped")
    n_mol = a["com"].shape[0]
    P = server_waves(n_mol)
    res = []
    for persistent in (1, 0):
        with make_batch(a, 4, persistent) as b:
            tot = b.potential_ewald(as_array=True)
            ch = b.new_chains(tot["energy"], tot["virial"])
            for n in (3 * n_mol, n_mol + 7):
                b.run_chains(ch, n, 298.15, seed=21, adjust=True, n_parts=P, n_threads=2)
            res.append((ch.copy(), state(b)))
    assert res[0][0].tobytes() == res[1][0].tobytes()
    assert same(res[0][1], res[1][1])
    assert len(set(res[0][0]["dr_max"])) > 1 or res[0][0]["dr_max"][0] != 0.15  # steps were adjusted


def test_server_default_is_automatic_and_uses_fewer_launches():
    """persistent = -1 (default): device_moves batches of up to 128 replicas take the server; its
    runs count one `launch` per step and group all the same (a control-word post)."""
    a = common.nist_arrays(1, "unwrapped")
    res = []
    for persistent in (-1, 1):
        with make_batch(a, 2, persistent, kernel=3) as b:
            e = b.potential_ewald(as_array=True)["energy"].copy()
            e, st = b.run(50, 298.15, 0.3, 0.05, 3, e, n_groups=1)
            assert st["server_steps"] == 50
            res.append((e, state(b)))
    assert np.array_equal(res[0][0], res[1][0]) and same(res[0][1], res[1][1])


@pytest.mark.parametrize("kernel,parts,wgs", [(2, 3, 0), (4, 8, -1)])
def test_server_sequence_numbers_wrap(kernel, parts, wgs):
    """The control word carries the step's sequence number in 24 bits.  A run that crosses 2^24
    (started 60 steps below it through the test hook; a single chain gets there after ~3 minutes)
    must go on, with both forms of the server, and sample the chain of the launch-per-step driver."""
    a = common.nist_arrays(1, "unwrapped")
    res = []
    for persistent in (1, 0):
        with make_batch(a, 2, persistent, kernel=kernel) as b:
            b.set_option("server_wgs", wgs)
            b.set_option("server_seq_offset", (1 << 24) - 60)
            e = b.potential_ewald(as_array=True)["energy"].copy()
            e, st = b.run(200, 298.15, 0.316555789, 0.05, 9, e, n_groups=1, n_parts=parts)
            assert st["server_steps"] == (200 if persistent else 0) and st["moves"] == 400
            res.append((e, state(b)))
    assert np.array_equal(res[0][0], res[1][0]) and same(res[0][1], res[1][1])


def test_server_refuses_what_it_cannot_do():
    a = common.nist_arrays(1, "unwrapped")
    with make_batch(a, 1, 1, kernel=0) as b:      # the generic kernel has no server form
        e = b.potential_ewald(as_array=True)["energy"].copy()
        with pytest.raises(MMCError, match="persistent move server needs"):
            b.run(3, 298.15, 0.3, 0.05, 3, e)
    with make_batch(a, 1, 1) as b:                 # host-side proposals
        b.set_option("device_moves", 0)
        e = b.potential_ewald(as_array=True)["energy"].copy()
        with pytest.raises(MMCError, match="persistent move server needs"):
            b.run(3, 298.15, 0.3, 0.05, 3, e)
        b.set_option("persistent", -1)             # automatic: falls back to launches
        assert b.run(3, 298.15, 0.3, 0.05, 3, e)[1]["server_steps"] == 0


def test_server_wait_is_bounded():
    """A host that stops talking for longer than the server waits (3 s): the workgroups give up and
    exit, the run reports it, nothing hangs, and the batch can be loaded again and used."""
    a = common.nist_arrays(1, "unwrapped")
    with make_batch(a, 2, 1) as b:
        e = b.potential_ewald(as_array=True)["energy"].copy()
        b.set_option("server_stall_ms", 3600)
        with pytest.raises(MMCError, match="timed out waiting for a control word"):
            b.run(10, 298.15, 0.3, 0.05, 3, e, n_groups=1)
        b.set_option("server_stall_ms", 0)
        # accepted moves, S-buffer parity and `e` may disagree now: the batch says so until every
        # replica has been loaded again
        with pytest.raises(MMCError, match="MMC_ERR_STATE"):
            b.run(10, 298.15, 0.3, 0.05, 3, e, n_groups=1)
        b.set_replica(0, a["com"], a["coords"])
        with pytest.raises(MMCError, match="MMC_ERR_STATE"):
            b.run(10, 298.15, 0.3, 0.05, 3, e, n_groups=1)
        for r in range(2):
            b.set_replica(r, a["com"], a["coords"])
        e0 = b.potential_ewald(as_array=True)["energy"].copy()
        assert np.array_equal(e0, e)
        e1, st = b.run(10, 298.15, 0.3, 0.05, 3, e0, n_groups=1)
        assert st["moves"] == 20


def test_latency_server_of_a_large_system():
    """10 000 molecules (BASELINE configs[3]): more than four workgroups' resident storage holds, so the
    server takes as many as it needs -- 21 workgroups, 84 parts -- and the host adds 21 records per
    step.  Bit for bit the chain of k_move_eval_lat launched per step with the same parts; the
    default shape (no part count given) is that server too."""
    from metropolismontecarlo_amd import io as mio, structs
    from metropolismontecarlo_amd.device import Batch
    nm = 10000
    box, com, coords = mio.cubic_lattice_water(nm, 0.033101144, "spce", seed=11234)
    a4 = common.nist_arrays(4, "unwrapped")
    atype = np.tile([1, 2, 2], nm)
    charge = np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], nm)
    res = []
    for persistent, parts in ((1, 84), (0, 84), (1, 0)):
        with Batch(1, com, coords, atype, charge, a4["eps"], a4["sig"], box, 5.6 / box, structs.factor,
                   10.0, 10.0) as b:
            b.set_option("device_moves", 1)
            b.set_option("persistent", persistent)
            if parts:
                b.set_option("kernel", 4)
            e = b.potential_ewald(as_array=True)["energy"].copy()
            e, st = b.run(120, 298.15, 0.3, 0.05, 23, e, n_groups=1, n_parts=parts)
            assert st["torn_records"] == 0 and st["server_steps"] == (120 if persistent else 0)
            e2 = b.potential_ewald(as_array=True)["energy"]
            assert np.abs(e - e2).max() <= 1e-11 * np.abs(e2).max()
            res.append((e.copy(), st["trans_accept"] + st["rot_accept"], state(b)))
    assert res[0][1] > 20
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0]) and res[0][1] == other[1] and same(res[0][2], other[2])
