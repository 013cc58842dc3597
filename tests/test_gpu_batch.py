"""GPU parity for the replica batch (mmc_batch_*): one fused launch per step for R replicas, the
commit-by-substitution of accepted moves, the S(k) buffer flip, and the native host driver.

Tolerance: TOL = 1e-9 relative to the per-molecule energy scale (north_star asks 1e-6).
"""
import numpy as np
import pytest

import common
from common import rel

pytestmark = pytest.mark.gpu

TOL = 1e-9
RCUT = 10.0


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def make_batch(a, R, **kw):
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Batch
    return Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
                 5.6 / a["box"], structs.factor, RCUT, RCUT, **kw)


def oracle_chain(orc, a, moves, accept_rule):
    """Run `moves` through the oracle with accept_rule(step, overlap) -> bool.  Returns the list
    of (d, overlap, accept) and the final (system, ewald)."""
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    orc.recip_long(ew, s.coords, s.charge, s.box)
    out = []
    for n, mv in enumerate(moves):
        i = mv["mol"]
        cn, an = np.array(mv["com_new"]), np.array(mv["atoms_new"])
        d, ov = orc.trial_move(i, s, ew, RCUT, RCUT, cn, an)
        acc = bool(accept_rule(n, ov)) and not ov
        if acc:
            s.com[i - 1] = cn
            s.coords[3 * (i - 1):3 * i] = an
            ew.sumQExpOld = ew.sumQExpNew.copy()
        else:
            ew.sumQExpNew = ew.sumQExpOld.copy()
        out.append((d, ov, acc))
    return out, s, ew


@pytest.mark.parametrize("kernel", [2, 1, 0])
@pytest.mark.parametrize("k,variant,parts", [(1, "reference", 1), (1, "reference", 4),
                                             (4, "reference", 0), (4, "unwrapped", 9),
                                             (2, "unwrapped", 2), (3, "reference", 16)])
def test_batch_eval_chain(k, variant, parts, kernel, orc):
    """Three replicas take the same proposals but different accept decisions, so their states
    diverge; every step's dU terms, overlap flags and the final device state (coordinates and
    S(k)) must match three independent oracle chains.  Proposals that touch the same molecule
    twice in a row and the 'previous move accepted' substitution are both exercised."""
    a = common.nist_arrays(k, variant)
    g = common.golden(k, variant)
    moves = list(g["moves"])
    # re-propose the molecule of an accepted move immediately (pending-commit == chosen molecule)
    extra = dict(moves[0])
    extra["com_new"] = (np.array(extra["com_new"]) + 0.05).tolist()
    extra["atoms_new"] = (np.array(extra["atoms_new"]) + 0.05).tolist()
    moves.insert(1, extra)
    rules = [lambda n, ov: True, lambda n, ov: False, lambda n, ov: n % 2 == 0]
    chains = [oracle_chain(orc, a, moves, r) for r in rules]
    R = len(rules)
    with make_batch(a, R) as b:
        if parts:
            b.set_parts(parts)
        b.set_option("kernel", kernel)   # 2: wave per move (default), 1: workgroup per move, 0: generic
        b.set_option("zero_copy_moves", k % 2)
        e0 = b.recip_long()
        ew = orc.Ewald(5.6 / a["box"], 5, 27, a["box"])
        assert rel(e0[0], orc.recip_long(ew, a["coords"], a["charge"], a["box"])) < TOL
        assert np.array_equal(e0, np.full(R, e0[0]))
        acc_prev = np.zeros(R, dtype=bool)
        for n, mv in enumerate(moves):
            d, ov = b.eval(mv["mol"], np.tile(mv["com_new"], (R, 1)),
                           np.tile(np.array(mv["atoms_new"]).ravel(), (R, 1)), acc_prev)
            for r in range(R):
                do, ovo, acco = chains[r][0][n]
                assert ov[r] == ovo, (n, r)
                scale = np.abs(do).max() + 1e4
                assert np.abs(d[r] - do).max() < TOL * scale, (n, r, d[r], do)
                acc_prev[r] = acco
        b.settle(acc_prev)
        for r in range(R):
            com, coords, S = b.get_replica(r)
            _, s, ewr = chains[r]
            assert np.array_equal(com, s.com) and np.array_equal(coords, s.coords), r
            assert np.abs(S - ewr.sumQExpOld).max() < 1e-11 * np.abs(ewr.sumQExpOld).max(), r
        tot = b.potential_ewald()
        for r in range(R):
            _, s, _ = chains[r]
            to = orc.potential_ewald(s, orc.Ewald(5.6 / s.box, 5, 27, s.box), RCUT, RCUT)
            for key in ("energy", "virial", "lj", "real", "recip", "self"):
                assert rel(tot[r][key], to[key]) < TOL, (r, key)


@pytest.mark.parametrize("k,variant,parts", [(1, "reference", 4), (1, "reference", 16),
                                             (4, "unwrapped", 16), (4, "reference", 8),
                                             (2, "unwrapped", 12), (3, "reference", 8)])
def test_batch_eval_chain_latency_kernel(k, variant, parts, orc):
    """test_batch_eval_chain for kernel 4 (k_move_eval_lat: resident molecule ranges, three lanes
    to a neighbour, the reciprocal sum split over waves, one record per workgroup): diverging
    replicas against three oracle chains, dU terms, overlap flags, final state."""
    a = common.nist_arrays(k, variant)
    moves = list(common.golden(k, variant)["moves"])
    extra = dict(moves[0])
    extra["com_new"] = (np.array(extra["com_new"]) + 0.05).tolist()
    extra["atoms_new"] = (np.array(extra["atoms_new"]) + 0.05).tolist()
    moves.insert(1, extra)
    rules = [lambda n, ov: True, lambda n, ov: False, lambda n, ov: n % 2 == 0]
    chains = [oracle_chain(orc, a, moves, r) for r in rules]
    R = len(rules)
    with make_batch(a, R) as b:
        b.set_option("kernel", 4)
        b.set_parts(parts)
        b.recip_long()
        acc_prev = np.zeros(R, dtype=bool)
        for n, mv in enumerate(moves):
            d, ov = b.eval(mv["mol"], np.tile(mv["com_new"], (R, 1)),
                           np.tile(np.array(mv["atoms_new"]).ravel(), (R, 1)), acc_prev)
            for r in range(R):
                do, ovo, acco = chains[r][0][n]
                assert ov[r] == ovo, (n, r)
                assert np.abs(d[r] - do).max() < TOL * (np.abs(do).max() + 1e4), (n, r, d[r], do)
                acc_prev[r] = acco
        b.settle(acc_prev)
        for r in range(R):
            com, coords, S = b.get_replica(r)
            _, s, ewr = chains[r]
            assert np.array_equal(com, s.com) and np.array_equal(coords, s.coords), r
            assert np.abs(S - ewr.sumQExpOld).max() < 1e-11 * np.abs(ewr.sumQExpOld).max(), r


def test_latency_kernel_refuses_ranges_it_cannot_hold():
    from metropolismontecarlo_amd._lib import MMCError
    a = common.nist_arrays(4, "unwrapped")            # 750 molecules: 4 parts = 3 ranges of 250
    mv = common.golden(4, "unwrapped")["moves"][0]
    with make_batch(a, 1) as b:
        b.set_option("kernel", 4)
        b.set_parts(4)
        b.recip_long()
        with pytest.raises(MMCError, match="kernel 4 needs"):
            b.eval(mv["mol"], np.tile(mv["com_new"], (1, 1)), np.tile(np.array(mv["atoms_new"]).ravel(), (1, 1)))


def test_batch_parts_agree(orc):
    """The split of a move over 1..16 workgroups changes only the summation order."""
    a = common.nist_arrays(4, "unwrapped")
    g = common.golden(4, "unwrapped")
    mv = g["moves"][0]
    ref = None
    for parts, kernel in ((1, 2), (2, 2), (1, 1), (2, 1), (3, 0), (5, 2), (9, 0), (16, 2),
                          (16, 1), (1, 0), (24, 2), (32, 1), (32, 0), (32, 2)):
        with make_batch(a, 2) as b:
            b.set_parts(parts)
            b.set_option("kernel", kernel)
            b.recip_long()
            d, ov = b.eval(mv["mol"], np.tile(mv["com_new"], (2, 1)),
                           np.tile(np.array(mv["atoms_new"]).ravel(), (2, 1)))
            assert np.array_equal(d[0], d[1])
            if ref is None:
                ref = d[0]
            assert np.abs(d[0] - ref).max() < 1e-9 * 1e4
            assert np.abs(d[0] - np.array(mv["d"])).max() < 1e-9 * 1e4


def test_batch_per_replica_configs(orc):
    """set_replica: replicas holding different configurations are evaluated independently."""
    a1 = common.nist_arrays(1, "reference")
    a2 = common.nist_arrays(1, "unwrapped")
    with make_batch(a1, 4) as b:
        b.set_replica(2, a2["com"], a2["coords"])
        t = b.potential_ewald()
        g1 = common.golden(1, "reference")["totals_ewald"]
        g2 = common.golden(1, "unwrapped")["totals_ewald"]
        for r, gg in ((0, g1), (1, g1), (2, g2), (3, g1)):
            assert rel(t[r]["energy"], gg["energy"]) < TOL and rel(t[r]["real"], gg["real"]) < TOL


def test_batch_errors():
    from metropolismontecarlo_amd._lib import MMCError
    a = common.nist_arrays(1)
    with make_batch(a, 2) as b:
        b.recip_long()
        with pytest.raises(MMCError, match="MMC_ERR_ARG"):
            b.eval(0, np.zeros((2, 3)), np.zeros((2, 9)))
        with pytest.raises(MMCError, match="MMC_ERR_ARG"):
            b.eval(101, np.zeros((2, 3)), np.zeros((2, 9)))
        b.eval(5, np.tile(a["com"][4], (2, 1)), np.tile(a["coords"][12:15].ravel(), (2, 1)))
        with pytest.raises(MMCError, match="MMC_ERR_STATE"):
            b.get_replica(0)          # proposals outstanding
        b.settle([0, 0])
        b.get_replica(0)
    with pytest.raises(AssertionError, match="nk == 5"):
        make_batch(a, 1, nk=4)
    bad = dict(a)
    bad["coords"] = a["coords"][:-1]
    with pytest.raises(AssertionError, match="n == 3"):
        make_batch(bad, 1)


def test_non_homogeneous_system_uses_generic_kernel(orc):
    """Different charges on different molecules: the fast kernel does not apply, the batch must
    fall back to the generic one by itself and refuse kernel=1."""
    from metropolismontecarlo_amd._lib import MMCError
    a = common.nist_arrays(1, "unwrapped")
    a = dict(a, charge=a["charge"].copy())
    a["charge"][3:6] *= 0.5          # molecule 2 is different (still neutral)
    g = common.golden(1, "unwrapped")
    mv = g["moves"][0]
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    orc.recip_long(ew, s.coords, s.charge, s.box)
    do, ovo = orc.trial_move(mv["mol"], s, ew, RCUT, RCUT, np.array(mv["com_new"]),
                             np.array(mv["atoms_new"]))
    with make_batch(a, 2) as b:
        with pytest.raises(MMCError, match="MMC_ERR_UNSUPPORTED"):
            b.set_option("kernel", 1)
        b.recip_long()
        d, ov = b.eval(mv["mol"], np.tile(mv["com_new"], (2, 1)),
                       np.tile(np.array(mv["atoms_new"]).ravel(), (2, 1)))
        assert ov[0] == ovo and np.abs(d[0] - do).max() < TOL * (np.abs(do).max() + 1e4)


@pytest.mark.parametrize("R,groups,parts,threads,kernel",
                         [(1, 1, 0, 1, 2), (5, 2, 0, 1, 2), (16, 3, 1, 3, 2), (8, 2, 4, 2, 0),
                          (12, 4, 0, 2, 1), (9, 2, 3, 2, 2)])
def test_engine_running_total_vs_recompute(R, groups, parts, threads, kernel, orc):
    """The reference's only integration invariant (Poly/main.jl:232-235): the running total
    energy (initial + accepted deltas) equals a full recompute -- here after hundreds of native
    driver steps, which also proves that commits reached the device coordinates and that the
    incrementally updated S(k) still matches a fresh RecipLong."""
    a = common.nist_arrays(1, "unwrapped")
    n_steps = 260  # > 2 sweeps of 100 molecules: every molecule is proposed several times
    with make_batch(a, R) as b:
        b.set_option("kernel", kernel)
        t0 = b.potential_ewald()
        e0 = np.array([t["energy"] for t in t0])
        e1, st = b.run(n_steps, 298.15, 0.316555789, 0.05, seed=11234, energies=e0,
                       n_groups=groups, n_parts=parts, n_threads=threads)
        assert st["moves"] == n_steps * R
        assert st["trans_attempt"] + st["rot_attempt"] == n_steps * R
        acc = st["trans_accept"] + st["rot_accept"]
        assert 0.2 * n_steps * R < acc < n_steps * R      # sane acceptance at 298 K
        S_inc = [b.get_replica(r)[2] for r in range(R)]
        t1 = b.potential_ewald()                          # refreshes S(k) from scratch
        for r in range(R):
            assert rel(e1[r], t1[r]["energy"]) < 1e-9, (r, e1[r], t1[r]["energy"])
            S_new = b.get_replica(r)[2]
            assert np.abs(S_inc[r] - S_new).max() < 1e-9 * np.abs(S_new).max()
        if R > 1:  # independent RNG streams: replicas must have diverged
            assert len({round(x, 6) for x in e1}) > 1
        # and the device state is what the oracle says about it
        com, coords, _ = b.get_replica(R - 1)
        s = common.oracle_system(dict(a, com=com, coords=coords))
        to = orc.potential_ewald(s, orc.Ewald(5.6 / s.box, 5, 27, s.box), RCUT, RCUT)
        assert rel(t1[R - 1]["energy"], to["energy"]) < TOL


def test_engine_deterministic_and_group_independent():
    a = common.nist_arrays(1, "unwrapped")
    res = []
    for groups, parts, threads in ((1, 1, 1), (3, 1, 3), (2, 5, 2)):
        with make_batch(a, 6) as b:
            b.recip_long()
            e, st = b.run(150, 298.15, 0.316555789, 0.05, seed=7, n_groups=groups, n_parts=parts,
                          n_threads=threads)
            res.append((e, st))
    # same workgroup split -> bitwise identical whatever the stream grouping
    assert np.array_equal(res[0][0], res[1][0])
    for key in ("trans_accept", "rot_accept", "overlaps"):
        assert res[0][1][key] == res[1][1][key]
    # different split: same chain unless a Metropolis draw lands within rounding of the threshold
    assert np.allclose(res[0][0], res[2][0], rtol=0, atol=1e-6)


def test_engine_molecules_stay_rigid():
    """Translations carry the atoms with the centre of mass through the periodic wrap and
    rotations are rigid: intramolecular distances are conserved over a run."""
    a = common.nist_arrays(2, "unwrapped")

    def bonds(c):
        c = c.reshape(-1, 3, 3)
        return np.stack([np.linalg.norm(c[:, 0] - c[:, 1], axis=1),
                         np.linalg.norm(c[:, 0] - c[:, 2], axis=1),
                         np.linalg.norm(c[:, 1] - c[:, 2], axis=1)])

    with make_batch(a, 2) as b:
        b.recip_long()
        b.run(600, 298.15, 0.316555789, 0.05, seed=3)
        com, coords, _ = b.get_replica(1)
        assert np.abs(bonds(coords) - bonds(a["coords"])).max() < 1e-9
        assert (com >= 0).all() and (com <= a["box"]).all()   # PBC (boundaries.jl:16-26)
        assert np.abs(coords - a["coords"]).max() > 1e-3        # something moved


def test_engine_chains_bookkeeping_and_adjust():
    """mmc_batch_run_chains: Loop()'s per-chain bookkeeping (main.jl:595-651).  Fixed step sizes
    reproduce mmc_batch_run bit for bit; the running virial matches a recompute; Adjust! after
    every sweep matches the host mirror of adjust.jl applied to the same counters."""
    from metropolismontecarlo_amd import moves
    from metropolismontecarlo_amd._lib import MMCError
    from metropolismontecarlo_amd.structs import Moves
    a = common.nist_arrays(1, "unwrapped")
    n_mol, R = a["com"].shape[0], 4
    with make_batch(a, R) as b:
        t0 = b.potential_ewald()
        e0 = np.array([t["energy"] for t in t0])
        v0 = np.array([t["virial"] for t in t0])
        e_plain, st_plain = b.run(2 * n_mol, 298.15, 0.3, 0.05, seed=5, energies=e0, n_threads=2)
    with make_batch(a, R) as b:
        b.potential_ewald()
        ch = b.new_chains(e0, v0, dr_max=0.3, dphi_max=0.05)
        st = b.run_chains(ch, 2 * n_mol, 298.15, seed=5, adjust=False, n_threads=2)
        assert np.array_equal(ch["energy"], e_plain)
        assert st["trans_accept"] == st_plain["trans_accept"] == ch["trans_naccept"].sum()
        assert (ch["steps_taken"] == 2 * n_mol).all() and (ch["dr_max"] == 0.3).all()
        assert (ch["trans_attempt"] + ch["rot_attempt"] == 2 * n_mol).all()
        assert ch["overlaps"].sum() == st["overlaps"]
        t1 = b.potential_ewald()
        for r in range(R):
            assert rel(ch["energy"][r], t1[r]["energy"]) < 1e-9
            assert rel(ch["virial"][r], t1[r]["virial"]) < 1e-9
        mean = ch["avg_energy"] / ch["steps_taken"]      # block average of the running total
        assert (mean < np.maximum(e0, ch["energy"]) + 5e4).all()
        assert (mean > np.minimum(e0, ch["energy"]) - 5e4).all()
        # averages.energy accumulates the running total after every move: one move at a time
        acc = ch["avg_energy"].copy()
        for k in range(6):
            b.run_chains(ch, 1, 298.15, seed=100 + k, adjust=False)
            acc += ch["energy"]
        assert np.allclose(ch["avg_energy"], acc, rtol=1e-14)

    with make_batch(a, R) as b:
        b.potential_ewald()
        ch = b.new_chains(e0, v0, dr_max=2.5, dphi_max=1.5, set_value=0.5)  # far too large
        mirror = [(Moves(0, 0, 0, 0, 0.5, 2.5), Moves(0, 0, 0, 0, 0.5, 1.5)) for _ in range(R)]
        for sweep in range(4):
            b.run_chains(ch, n_mol, 298.15, seed=50 + sweep, adjust=True, n_threads=3)
            for r in range(R):
                mt, mr = mirror[r]
                mt.naccept, mt.attempt = int(ch["trans_naccept"][r]), int(ch["trans_attempt"][r])
                mr.naccept, mr.attempt = int(ch["rot_naccept"][r]), int(ch["rot_attempt"][r])
                moves.Adjust(mt, a["box"])
                moves.Adjust_rot(mr, a["box"])
                assert ch["dr_max"][r] == mt.d_max and ch["dphi_max"][r] == mr.d_max
                assert ch["trans_attempp"][r] == mt.attempp and ch["rot_naccepp"][r] == mr.naccepp
        assert (ch["dr_max"] < 2.5).all()     # acceptance at 2.5 A is far below 50 %: shrunk
        # a partial sweep does not trigger the controller
        before = ch["dr_max"].copy()
        b.run_chains(ch, n_mol - 1, 298.15, seed=99, adjust=True)
        assert np.array_equal(ch["dr_max"], before)
        bad = ch.copy()
        bad["trans_set_value"][1] = 0.0
        with pytest.raises(MMCError):
            b.run_chains(bad, 1, 298.15, seed=1)


# ---- device-side move generation ------------------------------------------------
@pytest.mark.parametrize("R,groups,parts,threads,kernel,zero_copy",
                         [(1, 1, 0, 1, 2, 0), (7, 2, 0, 2, 2, 0), (16, 3, 1, 3, 2, 1),
                          (8, 2, 4, 2, 0, 0), (6, 2, 0, 2, 1, 0)])
def test_device_moves_running_total_vs_recompute(R, groups, parts, threads, kernel, zero_copy, orc):
    """Same invariant as test_engine_running_total_vs_recompute with the proposals drawn on the
    device: the move kernel, the commit and S(k) see exactly the coordinates k_propose wrote."""
    a = common.nist_arrays(1, "unwrapped")
    n_steps = 260
    with make_batch(a, R) as b:
        b.set_option("kernel", kernel)
        b.set_option("device_moves", 1)
        b.set_option("zero_copy_moves", zero_copy)
        e0 = np.array([t["energy"] for t in b.potential_ewald()])
        e1, st = b.run(n_steps, 298.15, 0.316555789, 0.05, seed=4242, energies=e0,
                       n_groups=groups, n_parts=parts, n_threads=threads)
        assert st["moves"] == n_steps * R
        assert st["trans_attempt"] + st["rot_attempt"] == n_steps * R
        assert 0.3 < st["trans_attempt"] / (n_steps * R) < 0.7      # chose_move < 0.5
        acc = st["trans_accept"] + st["rot_accept"]
        assert 0.2 * n_steps * R < acc < n_steps * R
        S_inc = [b.get_replica(r)[2] for r in range(R)]
        t1 = b.potential_ewald()
        for r in range(R):
            assert rel(e1[r], t1[r]["energy"]) < 1e-9, (r, e1[r], t1[r]["energy"])
            S_new = b.get_replica(r)[2]
            assert np.abs(S_inc[r] - S_new).max() < 1e-9 * np.abs(S_new).max()
        if R > 1:
            assert len({round(x, 6) for x in e1}) > 1
        com, coords, _ = b.get_replica(R - 1)
        s = common.oracle_system(dict(a, com=com, coords=coords))
        to = orc.potential_ewald(s, orc.Ewald(5.6 / s.box, 5, 27, s.box), RCUT, RCUT)
        assert rel(t1[R - 1]["energy"], to["energy"]) < TOL
        # back to host-side proposals on the same batch: the host mirror is refreshed first
        b.set_option("device_moves", 0)
        e2, _ = b.run(120, 298.15, 0.316555789, 0.05, seed=5, energies=e1, n_groups=groups,
                      n_threads=threads)
        t2 = b.potential_ewald()
        for r in range(R):
            assert rel(e2[r], t2[r]["energy"]) < 1e-9


def test_device_moves_distribution_of_one_step():
    """One step from a known state: an accepted translation displaces the COM by at most
    dr_max/2 per axis with the atoms following rigidly (auxillary.jl:94-103), an accepted
    rotation keeps the COM and turns the molecule by at most dphi_max (quaternions.jl:158-182)."""
    a = common.nist_arrays(2, "unwrapped")
    R, dr, dphi, box = 400, 0.3, 0.2, a["box"]
    with make_batch(a, R) as b:
        b.set_option("device_moves", 1)
        b.recip_long()
        # T -> infinity: Metropolis accepts (nearly) everything, so the accepted displacements
        # are the proposal distribution itself, not biased by the local energy gradient
        _, st = b.run(1, 1.0e12, dr, dphi, seed=77, n_threads=2)
        c0, x0 = a["com"][0], a["coords"][:3]
        n_t = n_r = 0
        zs, angles = [], []
        for r in range(R):
            com, coords, _ = b.get_replica(r)
            assert np.array_equal(com[1:], a["com"][1:]) and np.array_equal(coords[3:], a["coords"][3:])
            if np.array_equal(coords[:3], x0):
                continue                                  # rejected
            d = com[0] - c0
            if np.abs(d).max() > 0:                       # translation
                n_t += 1
                dw = d - box * np.round(d / box)
                assert np.abs(dw).max() <= dr / 2 + 1e-12
                assert (com[0] >= 0).all() and (com[0] <= box).all()
                assert np.abs((coords[:3] - x0) - d).max() < 1e-12
                zs.append(dw / dr)
            else:                                         # rotation about the COM
                n_r += 1
                o0, o1 = x0 - c0, coords[:3] - c0
                assert np.abs(np.linalg.norm(o1, axis=1) - np.linalg.norm(o0, axis=1)).max() < 1e-12
                # rotation angle from the trace of R, R = o1^T pinv(o0^T) on the molecular frame
                f0 = np.stack([o0[0], o0[1], np.cross(o0[0], o0[1])])
                f1 = np.stack([o1[0], o1[1], np.cross(o1[0], o1[1])])
                Rm = np.linalg.solve(f0, f1).T
                assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-9)
                ang = np.arccos(np.clip((np.trace(Rm) - 1) / 2, -1, 1))
                assert ang <= dphi + 1e-9
                angles.append(ang)
        assert n_t == st["trans_accept"] and n_r == st["rot_accept"]
        assert n_t > 150 and n_r > 150 and n_t + n_r >= R - 2
        zs = np.array(zs)                                  # uniform on (-1/2, 1/2): sigma of the
        assert abs(zs.mean()) < 0.06 and zs.min() < -0.45 and zs.max() > 0.45   # mean is 0.013
        assert abs(zs.std() - 12 ** -0.5) < 0.03
        assert max(angles) > 0.8 * dphi


def test_device_moves_deterministic_and_statistically_like_host_moves():
    a = common.nist_arrays(1, "unwrapped")
    res = []
    for groups, threads in ((1, 1), (3, 2)):
        with make_batch(a, 12) as b:
            b.set_option("device_moves", 1)
            b.recip_long()
            res.append(b.run(200, 298.15, 0.316555789, 0.05, seed=9, n_groups=groups, n_parts=1,
                             n_threads=threads))
    assert np.array_equal(res[0][0], res[1][0])       # counter-based draws: grouping cannot matter
    for key in ("trans_accept", "rot_accept", "trans_attempt", "overlaps"):
        assert res[0][1][key] == res[1][1][key]
    with make_batch(a, 12) as b:
        b.recip_long()
        _, host = b.run(200, 298.15, 0.316555789, 0.05, seed=9, n_parts=1)
    dev = res[0][1]
    ratio = lambda s: (s["trans_accept"] + s["rot_accept"]) / s["moves"]
    assert abs(ratio(dev) - ratio(host)) < 0.05


def test_device_moves_chains_adjust():
    a = common.nist_arrays(1, "unwrapped")
    n_mol, R = a["com"].shape[0], 3
    with make_batch(a, R) as b:
        b.set_option("device_moves", 1)
        t0 = b.potential_ewald()
        ch = b.new_chains([t["energy"] for t in t0], [t["virial"] for t in t0], dr_max=2.5,
                          dphi_max=1.5)
        b.run_chains(ch, 4 * n_mol, 298.15, seed=21, adjust=True, n_threads=2)
        assert (ch["dr_max"] < 1.5).all() and (ch["dphi_max"] < 1.5).all()
        assert (ch["steps_taken"] == 4 * n_mol).all()
        t1 = b.potential_ewald()
        for r in range(R):
            assert rel(ch["energy"][r], t1[r]["energy"]) < 1e-9
            assert rel(ch["virial"][r], t1[r]["virial"]) < 1e-9
        # the shrunken step sizes reached the device: every molecule moved less than the initial box
        com, coords, _ = b.get_replica(0)
        d = com - a["com"]
        d -= a["box"] * np.round(d / a["box"])
        assert np.abs(d).max() < 4 * 1.25 + 1e-9


def test_rdf_histogram_matches_gr_jl_restatement_bit_for_bit():
    """mmc_batch_rdf against the numpy restatement of gr.jl's pair loop: integer counts, exact."""
    from oracle import numpy_check
    from metropolismontecarlo_amd import observables
    a = common.nist_arrays(1, "unwrapped")
    n_mol, box, numbins = a["com"].shape[0], a["box"], 100
    with make_batch(a, 3) as b:
        b.recip_long()
        b.run(150, 298.15, 0.4, 0.2, seed=2, n_groups=1)          # replicas diverge
        want_o = np.zeros(numbins + 1, dtype=np.uint64)
        want_c = np.zeros(numbins + 1, dtype=np.uint64)
        for r in range(3):
            com, coords, _ = b.get_replica(r)
            want_o += numpy_check.make_rdf_hist(coords[0::3], box, numbins)
            want_c += numpy_check.make_rdf_hist(com, box, numbins)
        got_o, got_c = b.rdf(0, numbins), b.rdf(-1, numbins)
        assert np.array_equal(got_o, want_o) and np.array_equal(got_c, want_c)
        assert got_o.sum() > 0 and got_o[:10].sum() == 0           # no O-O closer than 1 A
        r_, g = observables.normalize_rdf(got_o, n_mol, box, 3)
        assert r_[0] == pytest.approx(0.5 * box / 2 / numbins)
        assert 1.5 < g.max() < 15.0 and 2.4 < r_[np.argmax(g)] < 3.2   # first O-O peak (noisy: 100 x 3)
        from metropolismontecarlo_amd._lib import MMCError
        with pytest.raises(MMCError, match="MMC_ERR_ARG"):
            b.rdf(0, 0)
        with pytest.raises(MMCError, match="MMC_ERR_ARG"):
            b.rdf(3, 10)


def test_fast_kernel_with_several_lj_pairs_per_molecule_pair(orc):
    """A homogeneous 3-site model whose three sites all carry LJ (9 LJ atom pairs per molecule
    pair, three atom types) runs the LDS-tiled kernel with n_ljp = 9: chain of moves against the
    oracle, both the one-workgroup and the split form."""
    rng = np.random.default_rng(31)
    a = common.random_system(150, 24.0, seed=31, na_choices=(3,), n_types=3)
    n_mol = a["com"].shape[0]
    a["atype"] = np.tile([1, 2, 3], n_mol).astype(np.int64)          # identical molecules
    q = np.array([-0.8, 0.5, 0.3])
    a["charge"] = np.tile(q, n_mol)
    e, s = np.array([60.0, 25.0, 8.0]), np.array([3.1, 2.6, 2.2])   # every pair has eps > 0.001
    a["eps"], a["sig"] = np.sqrt(e[:, None] * e[None, :]), (s[:, None] + s[None, :]) / 2
    rcut = 9.0
    for parts, kernel in ((1, 2), (3, 2), (1, 1), (3, 1)):
        s_o = common.oracle_system(a)
        ew = orc.Ewald(5.6 / a["box"], 5, 27, a["box"])
        orc.recip_long(ew, s_o.coords, s_o.charge, s_o.box)
        from metropolismontecarlo_amd import structs
        from metropolismontecarlo_amd.device import Batch
        with Batch(2, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
                   5.6 / a["box"], structs.factor, rcut, rcut) as b:
            b.set_option("kernel", kernel)                            # must qualify: homogeneous
            b.set_option("parts", parts)
            b.recip_long()
            acc_prev = None
            for step in range(12):
                i = int(rng.integers(1, n_mol + 1))
                d = (rng.random(3) - 0.5) * 0.4
                c_new = s_o.com[i - 1] + d
                a_new = s_o.coords[3 * (i - 1):3 * i] + d
                out, ov = b.eval(np.full(2, i), np.tile(c_new, (2, 1)), np.tile(a_new, (2, 1, 1)),
                                 accept_prev=acc_prev)
                do, ovo = orc.trial_move(i, s_o, ew, rcut, rcut, c_new, a_new)
                assert bool(ov[0]) == ovo and bool(ov[1]) == ovo
                assert np.abs(out[0] - do).max() < TOL * (np.abs(do).max() + 1e4), (step, out[0], do)
                assert np.array_equal(out[0], out[1])
                accept = (step % 3 != 0) and not ovo
                if accept:                                             # commit on both sides
                    s_o.com[i - 1] = c_new
                    s_o.coords[3 * (i - 1):3 * i] = a_new
                    ew.sumQExpOld[:] = ew.sumQExpNew
                else:
                    ew.sumQExpNew[:] = ew.sumQExpOld
                acc_prev = np.full(2, accept)
            b.settle(acc_prev)


@pytest.mark.parametrize("n_mol,R", [(101, 96), (751, 12), (75, 128)])
def test_half_pair_totals_with_an_odd_molecule_count(n_mol, R, orc):
    """k_total_wave in its PAIRED mode (a wave takes molecules u and N-1-u; chosen when the batch
    has at least 4096 such units) with an odd number of molecules: the middle molecule stands
    alone (`hasB = paired && iB > iA`).  All terms against the oracle, every replica the same."""
    a = _dense_water(n_mol, seed=3)
    assert (n_mol + 1) // 2 * R >= 4096 and n_mol % 2 == 1
    rcut = min(RCUT, a["box"] / 2)
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Batch
    with Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
               5.6 / a["box"], structs.factor, rcut, rcut) as b:
        t = b.potential_ewald(as_array=True)
    s = common.oracle_system(a)
    to = orc.potential_ewald(s, orc.Ewald(5.6 / s.box, 5, 27, s.box), rcut, rcut)
    for key in ("energy", "virial", "lj", "real", "recip", "self"):
        assert rel(t[key][0], to[key], 1.0) < TOL, key
        assert np.array_equal(t[key], np.full(R, t[key][0])), key
    assert t["n_overlap"][0] == to["n_overlap"]


def test_potential_ewald_record_array_equals_dict_list():
    a = common.nist_arrays(1, "unwrapped")
    with make_batch(a, 3) as b:
        dicts = b.potential_ewald()
        arr = b.potential_ewald(as_array=True)
        assert arr.shape == (3,)
        for r in range(3):
            for key in ("energy", "virial", "coulomb", "lj", "real", "recip", "self", "n_overlap"):
                assert arr[key][r] == dicts[r][key]


# ---- the bench's own path at its own size, and the loops small systems never reach ----------------
@pytest.mark.parametrize("kernel", [2, 1])
def test_bench_path_750_molecules_device_moves(kernel, orc):
    """What bench.py runs, under the parity gate: mmc_batch_run with device-side proposals
    (k_propose + k_fetch_bytes), the per-move kernel with one part per move, NIST configuration 4
    (750 molecules), two groups, more than two sweeps.  Running totals against a recompute
    (Poly/main.jl:232-235) for every replica, and the final device state of two replicas against
    the oracle's total energy of those coordinates."""
    a = common.nist_arrays(4, "unwrapped")
    n_mol, R = a["com"].shape[0], 64
    n_steps = 2 * n_mol + 37
    with make_batch(a, R) as b:
        b.set_option("kernel", kernel)
        b.set_option("device_moves", 1)
        e0 = b.potential_ewald(as_array=True)["energy"].copy()
        e1, st = b.run(n_steps, 298.15, 0.316555789, 0.05, seed=11234, energies=e0, n_groups=2,
                       n_parts=1, n_threads=2)
        # (where the wave kernel takes the decisions itself it takes eight steps per launch; 64 chains
        # run on the move server, which counts a launch per step and group)
        assert st["moves"] == n_steps * R
        assert st["launches"] == 2 * (-(-n_steps // 8) if st["device_decisions"] else n_steps)
        assert st["torn_records"] == 0
        acc = (st["trans_accept"] + st["rot_accept"]) / st["moves"]
        assert 0.6 < acc < 0.9, acc                 # 0.756 in the bench's runs
        S_inc = [b.get_replica(r)[2] for r in (0, R - 1)]
        t1 = b.potential_ewald(as_array=True)
        assert np.abs(e1 - t1["energy"]).max() < 1e-9 * np.abs(t1["energy"]).max()
        assert len({round(x, 6) for x in e1}) == R  # every chain went its own way
        for k, r in enumerate((0, R - 1)):
            com, coords, S_new = b.get_replica(r)
            assert np.abs(S_inc[k] - S_new).max() < 1e-9 * np.abs(S_new).max()
            s = common.oracle_system(dict(a, com=com, coords=coords))
            to = orc.potential_ewald(s, orc.Ewald(5.6 / s.box, 5, 27, s.box), RCUT, RCUT)
            for key in ("energy", "lj", "real", "recip"):
                assert rel(t1[key][r], to[key]) < TOL, (r, key)
            assert (com >= 0).all() and (com <= a["box"]).all()
            assert np.abs(com - a["com"]).max() > 0.05          # it moved


def _rigid_proposal(seed, replica, step, com, atoms, box, dr_max, dphi_max):
    """The move k_propose's rigid generator makes for (seed, replica, step) of a molecule at
    com / atoms, rebuilt on the host from the exported Philox draws with the reference's
    formulas (see test_gpu_moves.py).  Returns (kind, com_new, atoms_new, metropolis uniform)."""
    import math
    from metropolismontecarlo_amd import moves
    from test_gpu_moves import ReferenceOrder, SLOT_METROPOLIS, philox_pair
    draws = ReferenceOrder(seed, replica, step)
    u_met = philox_pair(seed, replica, step, SLOT_METROPOLIS)[0]
    if draws.chose_move() < 0.5:                                           # main.jl:519
        draws.start_translation()
        c_new = moves.random_translate_vector(dr_max, com, box, draws)
        return 0, c_new, atoms + (c_new - com), u_met
    draws.start_rotation()
    axis = moves.random_vector(draws)
    angle = (2.0 * draws.random() - 1.0) * dphi_max
    c, sn = math.cos(angle), math.sin(angle)
    t = 1.0 - c
    ex, ey, ez = axis
    Rm = np.array([[t * ex * ex + c, t * ex * ey - sn * ez, t * ex * ez + sn * ey],
                   [t * ex * ey + sn * ez, t * ey * ey + c, t * ey * ez - sn * ex],
                   [t * ex * ez - sn * ey, t * ey * ez + sn * ex, t * ez * ez + c]])
    return 1, com.copy(), com + (atoms - com) @ Rm.T, u_met


@pytest.mark.parametrize("kernel,R,parts,on_device", [(2, 24, 1, 0), (2, 24, 1, 1), (1, 2, 0, 0), (4, 1, 16, 0)])
def test_driver_device_moves_stepped_by_the_oracle(kernel, R, parts, on_device, orc):
    """mmc_batch_run with device-side proposals at 750 molecules, every step checked: the oracle
    steps the same chains -- proposal rebuilt from the exported Philox draws, dU from
    orc.trial_move, Metropolis (auxillary.jl:106-114) with the step's own uniform -- and the
    driver's recorded dU and decision of EVERY step (option "trace_steps"), accepted or
    rejected, must agree: a wrong dU on a rejected move cannot hide in the final state.
    on_device = 1: the decisions are the move kernel's own (option "accept_on_device")."""
    import math
    a = common.nist_arrays(4, "unwrapped")
    n_mol, box = a["com"].shape[0], a["box"]
    n_steps, seed, T, dr, dphi = 90, 424242, 298.15, 0.316555789, 0.05
    check = (0, R - 1) if R > 1 else (0,)
    with make_batch(a, R) as b:
        b.set_option("kernel", kernel)
        b.set_option("device_moves", 1)
        b.set_option("persistent", 0 if kernel != 4 else 1)   # launch per step; the latency server
        b.set_option("accept_on_device", on_device)
        b.set_option("trace_steps", n_steps)
        e0 = b.potential_ewald(as_array=True)["energy"].copy()
        e1, st = b.run(n_steps, T, dr, dphi, seed=seed, energies=e0, n_groups=min(R, 2),
                       n_parts=parts, n_threads=2, replica0=5)
        assert st["device_decisions"] == (R * n_steps if on_device else 0)
        d_gpu, f_gpu = b.get_trace(n_steps)
        final = {r: b.get_replica(r) for r in check}
    n_rej = n_rot = 0
    for r in check:
        s = common.oracle_system(a)
        ew = orc.Ewald(5.6 / box, 5, 27, box)
        orc.recip_long(ew, s.coords, s.charge, box)
        running = 0.0
        for step in range(n_steps):
            i = step % n_mol                                              # main.jl:490
            kind, c_new, a_new, u = _rigid_proposal(seed, 5 + r, step, s.com[i].copy(),
                                                    s.coords[3 * i:3 * i + 3].copy(), box, dr, dphi)
            d, ov = orc.trial_move(i + 1, s, ew, RCUT, RCUT, c_new, a_new)
            delta = d[0] + d[1] + d[2]                                    # main.jl:593
            x = delta / T
            accept = (x < 0.0 or math.exp(-x) > u) and not ov             # main.jl:598
            assert abs(d_gpu[r, step] - delta) < TOL * (abs(delta) + 1e4), (r, step, kind)
            assert f_gpu[r, step] == (int(accept) | (int(ov) << 1) | (kind << 2)), (r, step)
            n_rej += not accept
            n_rot += kind
            if accept:
                running += delta
                s.com[i] = c_new
                s.coords[3 * i:3 * i + 3] = a_new
                ew.sumQExpOld = ew.sumQExpNew.copy()
            else:
                ew.sumQExpNew = ew.sumQExpOld.copy()
        com, coords, S = final[r]
        assert np.abs(com - s.com).max() < 2e-13 and np.abs(coords - s.coords).max() < 2e-13
        assert np.abs(S - ew.sumQExpOld).max() < 1e-11 * np.abs(ew.sumQExpOld).max()
        assert abs((e1[r] - e0[r]) - running) < TOL * 1e5
    assert n_rej > 5 and n_rot > 10          # rejected moves and rotations were among the checked


@pytest.mark.parametrize("per_launch,n_steps", [(8, 90), (16, 41)])
def test_several_steps_per_launch_stepped_by_the_oracle(per_launch, n_steps, orc):
    """The headline's mode against the ORACLE: the move kernel decides and one launch takes a replica
    through several steps; no record of single steps comes back, so the oracle steps the same chains
    on its own -- proposals rebuilt from the Philox draws, dU from orc.trial_move, Metropolis
    (auxillary.jl:106-114) with math.exp and the step's uniform -- and the batch must end where the
    oracle does: accept counts, the sum of the accepted dU, coordinates, S(k)."""
    import math
    a = common.nist_arrays(4, "unwrapped")
    n_mol, box = a["com"].shape[0], a["box"]
    R, seed, T, dr, dphi = 24, 777, 298.15, 0.316555789, 0.05
    check = (0, 11, R - 1)
    with make_batch(a, R) as b:
        b.set_option("kernel", 2)
        b.set_option("device_moves", 1)
        b.set_option("persistent", 0)
        b.set_option("accept_on_device", 1)
        b.set_option("steps_per_launch", per_launch)
        e0 = b.potential_ewald(as_array=True)["energy"].copy()
        e1, st = b.run(n_steps, T, dr, dphi, seed=seed, energies=e0, n_groups=2, n_parts=1, n_threads=2,
                       replica0=3)
        assert st["device_decisions"] == R * n_steps and st["launches"] == 2 * -(-n_steps // per_launch)
        final = {r: b.get_replica(r) for r in check}
    for r in check:
        s = common.oracle_system(a)
        ew = orc.Ewald(5.6 / box, 5, 27, box)
        orc.recip_long(ew, s.coords, s.charge, box)
        running, n_acc = 0.0, 0
        for step in range(n_steps):
            i = step % n_mol
            kind, c_new, a_new, u = _rigid_proposal(seed, 3 + r, step, s.com[i].copy(),
                                                    s.coords[3 * i:3 * i + 3].copy(), box, dr, dphi)
            d, ov = orc.trial_move(i + 1, s, ew, RCUT, RCUT, c_new, a_new)
            delta = d[0] + d[1] + d[2]
            x = delta / T
            if (x < 0.0 or math.exp(-x) > u) and not ov:
                running += delta
                n_acc += 1
                s.com[i] = c_new
                s.coords[3 * i:3 * i + 3] = a_new
                ew.sumQExpOld = ew.sumQExpNew.copy()
            else:
                ew.sumQExpNew = ew.sumQExpOld.copy()
        com, coords, S = final[r]
        assert np.abs(com - s.com).max() < 2e-13 and np.abs(coords - s.coords).max() < 2e-13, r
        assert np.abs(S - ew.sumQExpOld).max() < 1e-11 * np.abs(ew.sumQExpOld).max(), r
        assert abs((e1[r] - e0[r]) - running) < TOL * 1e5, r
        assert 0 < n_acc < n_steps


@pytest.mark.parametrize("cfg,parts", [(4, 1), (4, 3), (1, 1)])
def test_minimum_image_by_molecule_is_bit_identical(cfg, parts):
    """Option "image_by_molecule": where gate + 2 r_mol < box / 2 (NIST configuration 4: 750 molecules
    in 30 A) the wave kernel takes an atom pair's minimum image from its molecule's -- and must give
    bit for bit what the per-pair minimum image gives, step by step; where the condition fails
    (configuration 1: 20 A) the option changes nothing because the kernel is not taken."""
    a = common.nist_arrays(cfg, "unwrapped")
    n_steps, seed, T, dr, dphi = 64, 99, 298.15, 0.316555789, 0.05
    out = []
    for opt in (-1, 0):
        with make_batch(a, 24) as b:
            b.set_option("kernel", 2)
            b.set_option("device_moves", 1)
            b.set_option("persistent", 0)
            b.set_option("image_by_molecule", opt)
            b.set_option("trace_steps", n_steps)
            e0 = b.potential_ewald(as_array=True)["energy"].copy()
            e1, st = b.run(n_steps, T, dr, dphi, seed=seed, energies=e0, n_groups=2, n_parts=parts,
                           n_threads=2)
            d, f = b.get_trace(n_steps)
            out.append((e1.copy(), d.copy(), f.copy(), st["trans_accept"] + st["rot_accept"]))
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.array_equal(out[0][0], out[1][0]) and out[0][3] == out[1][3]
    assert 0 < out[0][3] < 24 * n_steps


def _dense_water(n_mol, seed=5):
    from metropolismontecarlo_amd import io as mio
    box, com, coords = mio.cubic_lattice_water(n_mol, 0.033101144, "spce", seed=seed)
    a4 = common.nist_arrays(4, "unwrapped")
    first = 3 * np.arange(n_mol, dtype=np.int64) + 1
    return dict(com=com, coords=coords, first_atom=first, last_atom=first + 2,
                atype=np.tile([1, 2, 2], n_mol).astype(np.int64),
                charge=np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], n_mol), eps=a4["eps"],
                sig=a4["sig"], box=float(box))


def test_table_kernels_are_refused_beyond_16_bit_molecule_indices():
    """The wave kernels keep neighbour lists of 16-bit molecule indices: a system of more than 65535
    molecules gets the generic kernel, and asking for another one is an error, not a wrong answer."""
    from metropolismontecarlo_amd import structs, _lib
    from metropolismontecarlo_amd.device import Batch
    a = _dense_water(65600)
    with Batch(1, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
               5.6 / a["box"], structs.factor, 10.0, 10.0) as b:
        for k in (1, 2, 3, 4):
            with pytest.raises(_lib.MMCError) as e:
                b.set_option("kernel", k)
            assert "65535" in str(e.value)
        b.set_option("kernel", 0)


@pytest.mark.parametrize("kernel,parts", [(2, 1), (1, 1), (2, 3), (1, 2)])
def test_long_neighbour_lists_and_many_molecules(kernel, parts, orc):
    """1700 molecules with a 12.4 A cutoff: ~265 neighbours inside the gate, i.e. more than one
    neighbour tile (150) and more than one molecule chunk (768) for the workgroup-per-move kernel,
    and a neighbour list that overflows its 256 slots mid-scan for the wave-per-move kernel.
    A chain of moves against the oracle and against the generic kernel, then a native-driver run
    whose running total must match a recompute."""
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Batch
    a = _dense_water(1700)
    n_mol, rcut = 1700, 12.4
    s_o = common.oracle_system(a)
    ew = orc.Ewald(5.6 / a["box"], 5, 27, a["box"])
    orc.recip_long(ew, s_o.coords, s_o.charge, s_o.box)
    rng = np.random.default_rng(17)

    def mk():
        return Batch(2, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
                     5.6 / a["box"], structs.factor, rcut, rcut)

    with mk() as b, mk() as bg:
        b.set_option("kernel", kernel)
        b.set_option("parts", parts)
        bg.set_option("kernel", 0)
        bg.set_option("parts", 1)
        b.recip_long()
        bg.recip_long()
        acc_prev = None
        for step in range(6):
            i = int(rng.integers(1, n_mol + 1)) if step != 3 else 1700   # the last molecule too
            d = (rng.random(3) - 0.5) * 0.5
            c_new = s_o.com[i - 1] + d
            a_new = s_o.coords[3 * (i - 1):3 * i] + d
            out, ov = b.eval(np.full(2, i), np.tile(c_new, (2, 1)), np.tile(a_new, (2, 1, 1)),
                             accept_prev=acc_prev)
            outg, ovg = bg.eval(np.full(2, i), np.tile(c_new, (2, 1)), np.tile(a_new, (2, 1, 1)),
                                accept_prev=acc_prev)
            do, ovo = orc.trial_move(i, s_o, ew, rcut, rcut, c_new, a_new)
            scale = np.abs(do).max() + 1e4
            assert bool(ov[0]) == ovo == bool(ovg[0])
            assert np.abs(out[0] - do).max() < TOL * scale, (step, out[0], do)
            assert np.abs(out[0] - outg[0]).max() < TOL * scale
            assert np.array_equal(out[0], out[1])
            accept = (step % 2 == 0) and not ovo
            if accept:
                s_o.com[i - 1] = c_new
                s_o.coords[3 * (i - 1):3 * i] = a_new
                ew.sumQExpOld[:] = ew.sumQExpNew
            else:
                ew.sumQExpNew[:] = ew.sumQExpOld
            acc_prev = np.full(2, accept)
        b.settle(acc_prev)
        bg.settle(acc_prev)
        for dev in (0, 1):
            b.set_option("device_moves", dev)
            e0 = b.potential_ewald(as_array=True)["energy"].copy()
            e1, st = b.run(60, 298.15, 0.3, 0.05, seed=3, energies=e0, n_groups=1, n_parts=parts)
            t1 = b.potential_ewald(as_array=True)["energy"]
            assert np.abs(e1 - t1).max() < 1e-9 * np.abs(t1).max()
            assert st["trans_accept"] + st["rot_accept"] > 0


def test_config3_one_workgroup_per_move(orc):
    """NIST configuration 3 (300 molecules in a 20 A box, ~157 neighbours): the second neighbour
    tile of the workgroup-per-move kernel with the whole move in one workgroup.  Every proposal is
    rejected, so each is evaluated from the initial state -- on all three kernels and the oracle."""
    a = common.nist_arrays(3, "unwrapped")
    g = common.golden(3, "unwrapped")
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    orc.recip_long(ew, s.coords, s.charge, s.box)
    want = []
    for mv in g["moves"][:4]:
        do, ovo = orc.trial_move(mv["mol"], s, ew, RCUT, RCUT, np.array(mv["com_new"]),
                                 np.array(mv["atoms_new"]))
        ew.sumQExpNew[:] = ew.sumQExpOld
        want.append((do, ovo))
    for kernel in (0, 1, 2):
        with make_batch(a, 2) as b:
            b.set_option("kernel", kernel)
            b.set_option("parts", 1)
            b.recip_long()
            for mv, (do, ovo) in zip(g["moves"][:4], want):
                d, ov = b.eval(mv["mol"], np.tile(mv["com_new"], (2, 1)),
                               np.tile(np.array(mv["atoms_new"]).ravel(), (2, 1)))
                assert bool(ov[0]) == ovo
                assert np.abs(d[0] - do).max() < TOL * (np.abs(do).max() + 1e4), (kernel, d[0], do)
            b.settle([0, 0])


# ---- the result hand-off: a record is consumed only if stamp and checksum agree -------------------
def test_part_record_checksum_refuses_a_torn_record():
    a = common.nist_arrays(1, "unwrapped")
    mv = common.golden(1, "unwrapped")["moves"][0]
    for kernel, parts in ((2, 1), (2, 3), (1, 2), (0, 1)):
        with make_batch(a, 2) as b:
            b.set_option("kernel", kernel)
            b.set_option("parts", parts)
            b.recip_long()
            b.eval(mv["mol"], np.tile(mv["com_new"], (2, 1)),
                   np.tile(np.array(mv["atoms_new"]).ravel(), (2, 1)))
            for part in range(parts):
                raw, stamp = b.peek_part(1, part)
                assert len(raw) == 64 and b.part_validate(raw, stamp)
                assert not b.part_validate(raw, stamp + 1)            # a stale launch's record
                for piece in range(4):   # one 16-byte piece left over from another launch
                    torn = bytearray(raw)
                    for k in range(16):
                        torn[16 * piece + k] ^= 0x5A if piece < 3 or k < 8 else 0
                    if piece == 3:       # keep the stamp word, corrupt the recip sum next to it
                        assert bytes(torn[56:64]) == raw[56:64]
                    assert not b.part_validate(bytes(torn), stamp), (kernel, part, piece)
                flipped = bytearray(raw)
                flipped[20] ^= 1          # a single bit of one sum
                assert not b.part_validate(bytes(flipped), stamp)
            b.settle([0, 0])


@pytest.mark.parametrize("device_moves", [1, 0])
def test_driver_rereads_torn_records(device_moves):
    """inject_torn: the driver's first N result copies are corrupted the way a torn PCIe write
    would look (new stamp, one 16-byte piece stale).  It must refuse each of them, read again,
    and arrive at exactly the chain an undisturbed run produces."""
    a = common.nist_arrays(1, "unwrapped")
    res = []
    for inject in (0, 25):
        with make_batch(a, 6) as b:
            b.set_option("device_moves", device_moves)
            b.recip_long()
            if inject:
                b.set_option("inject_torn", inject)
            e, st = b.run(40, 298.15, 0.316555789, 0.05, seed=9, n_groups=2, n_threads=2)
            res.append((e, st))
    assert res[0][1]["torn_records"] == 0 and res[1][1]["torn_records"] == 25
    assert np.array_equal(res[0][0], res[1][0])
    for key in ("trans_accept", "rot_accept", "overlaps", "moves"):
        assert res[0][1][key] == res[1][1][key]


@pytest.mark.parametrize("k", [1, 4])
def test_total_energy_kernels_of_large_batches(k, orc):
    """R = 256 takes the many-replica forms of the total-energy kernels (a wave per pair of
    molecules for the pair part; for the structure factor a workgroup per replica with every
    atom's phases in LDS -- 17 KB at 100 molecules, 126 KB of the 160 KB at 750): replicas
    holding three different configurations against the oracle, terms and S(k)."""
    a1 = common.nist_arrays(k, "unwrapped")
    a2 = common.nist_arrays(k, "reference")
    rng = np.random.default_rng(3)
    shifted = dict(a1, com=a1["com"].copy(), coords=a1["coords"].copy())
    d = (rng.random(3) - 0.5) * 0.4
    shifted["com"][17] += d
    shifted["coords"][51:54] += d
    R = 256
    with make_batch(a1, R) as b:
        b.set_replica(100, a2["com"], a2["coords"])
        b.set_replica(R - 1, shifted["com"], shifted["coords"])
        t = b.potential_ewald(as_array=True)
        for r, a in ((0, a1), (100, a2), (R - 1, shifted), (R - 2, a1)):
            s = common.oracle_system(a)
            ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
            to = orc.potential_ewald(s, ew, RCUT, RCUT)
            for key in ("energy", "virial", "lj", "real", "recip", "self"):
                assert rel(t[key][r], to[key]) < TOL, (r, key, t[key][r], to[key])
            S = b.get_replica(r)[2]
            assert np.abs(S - ew.sumQExpNew).max() < 1e-11 * np.abs(ew.sumQExpNew).max(), r
        assert np.array_equal(t["energy"][1:100], np.full(99, t["energy"][0]))


@pytest.mark.parametrize("groups,threads", [(2, 2), (4, 3), (3, 1)])
def test_stream_layout_does_not_change_the_chains(groups, threads):
    """The driver's default with device-side proposals spreads the replica groups over two streams
    (their launches overlap) and launches the first group's first step before its worker threads
    exist; one stream runs every launch alone.  Chains are per-replica Markov chains: energies,
    statistics, coordinates and S(k) must be IDENTICAL bit for bit whatever the layout, over
    two calls in a row (the second continues streams, S-buffer parity and the proposal ring)."""
    a = common.nist_arrays(4, "unwrapped")
    R = 96
    res = []
    for n_streams in (0, 1, groups):
        with make_batch(a, R) as b:
            b.set_option("kernel", 2)
            b.set_option("device_moves", 1)
            b.set_option("persistent", 0)
            e = b.potential_ewald(as_array=True)["energy"].copy()
            stats = []
            for n in (41, 23):
                e, st = b.run(n, 298.15, 0.316555789, 0.05, seed=77, energies=e, n_groups=groups,
                              n_parts=1, n_threads=threads, n_streams=n_streams)
                stats.append([st[q] for q in ("moves", "trans_accept", "rot_accept", "overlaps", "launches")])
            assert st["torn_records"] == 0
            res.append((e.copy(), stats, [b.get_replica(r) for r in (0, R // 2, R - 1)]))
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0]) and res[0][1] == other[1]
        for x, y in zip(res[0][2], other[2]):
            assert all(np.array_equal(p, q) for p, q in zip(x, y))
    assert res[0][1][0][0] == 41 * R and res[0][1][0][4] == -(-41 // 8) * groups   # eight steps per launch


def test_half_space_k_list_gives_the_reference_structure_factor(orc):
    """A batch keeps 293 of the reference's 337 k-vectors: of each conjugate pair (0, ky, kz) /
    (0, -ky, -kz) the first, with twice the weight (k_kvec_setup).  mmc_batch_get_replica hands
    the reference's array back; checked here against the oracle's sumQExp -- all 337 entries --
    and against the same batch run with the full list (MMC_FULL_K): the same chains (accept
    counts, coordinates, S(k) bit for bit), energies that differ by the order of a 337- / 293-term
    sum only."""
    import os
    a = common.nist_arrays(4, "unwrapped")
    R = 64
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    orc.recip_long(ew, s.coords, s.charge, s.box)
    kx0 = [i for i, v in enumerate(ew.kxyz.tolist()) if v[0] == 0]
    assert len(ew.kxyz) == 337 and len(kx0) == 88
    out = {}
    for full in (False, True):
        if full:
            os.environ["MMC_FULL_K"] = "1"
        try:
            with make_batch(a, R) as b:
                b.set_option("kernel", 2)
                b.set_option("device_moves", 1)
                t0 = b.potential_ewald(as_array=True)
                S0 = b.get_replica(R - 1)[2]
                e, st = b.run(60, 298.15, 0.316555789, 0.05, seed=5, energies=t0["energy"].copy(),
                              n_groups=2, n_parts=1, n_threads=2)
                out[full] = (t0["recip"].copy(), S0, e.copy(),
                             [st[q] for q in ("moves", "trans_accept", "rot_accept", "overlaps")],
                             [b.get_replica(r) for r in (0, R - 1)], b.potential_ewald(as_array=True)["energy"].copy())
        finally:
            os.environ.pop("MMC_FULL_K", None)
    half, full = out[False], out[True]
    assert np.abs(half[1] - ew.sumQExpNew).max() < 1e-11 * np.abs(ew.sumQExpNew).max()
    assert np.array_equal(half[1], full[1])                    # S(k) after RecipLong, 337 entries
    to = orc.potential_ewald(common.oracle_system(a), orc.Ewald(5.6 / s.box, 5, 27, s.box), RCUT, RCUT)
    assert rel(half[0][0], to["recip"]) < 1e-12
    assert np.abs(half[0] - full[0]).max() < 1e-13 * np.abs(full[0]).max()
    assert half[3] == full[3] and half[3][0] == 60 * R         # the same chains
    for x, y in zip(half[4], full[4]):
        assert all(np.array_equal(p, q) for p, q in zip(x, y))
        S = x[2]
        pairs = {tuple(v): i for i, v in enumerate(ew.kxyz.tolist())}
        for i in kx0:                                          # exact conjugates, as in the reference
            j = pairs[(0, -ew.kxyz[i][1], -ew.kxyz[i][2])]
            assert S[j].real == S[i].real and S[j].imag == -S[i].imag
    assert np.abs(half[2] - full[2]).max() < 1e-12 * np.abs(full[2]).max()
    assert np.abs(half[2] - half[5]).max() < 1e-11 * np.abs(half[5]).max()   # running = recomputed


@pytest.mark.parametrize("groups,threads,calls", [(2, 2, (41, 23, 1, 2)), (1, 1, (7, 12)), (3, 3, (30, 5, 17))])
def test_accept_decision_in_the_kernel_gives_the_same_chains(groups, threads, calls):
    """Option "accept_on_device": the move kernel takes the Metropolis decision itself (the same
    Philox uniform, dU in the host's arithmetic), keeps accept flag and S-buffer bit in device
    memory and sends the decision along in its result record; the host only keeps the books.  With
    "steps_per_launch" > 1 the same wave takes a replica through several steps per launch and sends
    one record for all of them.  The chains must be the host-decided ones bit for bit -- counts,
    coordinates, S(k); the energies too with one step per launch, and to the order of a sum with
    several -- over several calls in a row (step counts that are and are not multiples of the steps
    per launch; a call of one step), for one and several groups, and a torn result record must still
    be caught."""
    a = common.nist_arrays(4, "unwrapped")
    R = 96
    res = []
    modes = [(0, 1, 0), (1, 1, 0), (1, 1, 5), (1, 2, 0), (1, 8, 0), (1, 16, 3), (-1, 0, 0)]
    for on_device, per_launch, torn_in in modes:
        with make_batch(a, R) as b:
            b.set_option("kernel", 2)
            b.set_option("device_moves", 1)
            b.set_option("persistent", 0)
            b.set_option("accept_on_device", on_device)
            b.set_option("steps_per_launch", per_launch)
            if torn_in:
                b.set_option("inject_torn", torn_in)
            e = b.potential_ewald(as_array=True)["energy"].copy()
            stats, torn, launches = [], [], []
            for n in calls:
                e, st = b.run(n, 298.15, 0.316555789, 0.05, seed=31, energies=e, n_groups=groups,
                              n_parts=1, n_threads=threads)
                assert st["device_decisions"] == (n * R if on_device else 0)
                stats.append([st[q] for q in ("moves", "trans_attempt", "trans_accept", "rot_attempt",
                                              "rot_accept", "overlaps")])
                launches.append(st["launches"])
                torn.append(st["torn_records"])
            assert torn == [torn_in] + [0] * (len(calls) - 1)
            k = per_launch if per_launch else 8                 # (0: by size -- eight here)
            assert launches == [-(-n // k) * groups for n in calls]
            res.append((e.copy(), stats, [b.get_replica(r) for r in (0, R // 2, R - 1)],
                        b.potential_ewald(as_array=True)["energy"].copy()))
    for (on_device, per_launch, _), other in zip(modes[1:], res[1:]):
        assert res[0][1] == other[1]
        if per_launch == 1:
            assert np.array_equal(res[0][0], other[0])
        assert np.abs(res[0][0] - other[0]).max() < 1e-12 * np.abs(res[0][0]).max()
        for x, y in zip(res[0][2], other[2]):
            assert all(np.array_equal(p, q) for p, q in zip(x, y))
    assert 0 < sum(c[2] + c[4] for c in res[0][1]) < sum(calls) * R
    for q in res:
        assert np.abs(q[0] - q[3]).max() < 1e-11 * np.abs(q[3]).max()   # running = recomputed
