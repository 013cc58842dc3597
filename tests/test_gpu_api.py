"""GPU parity through the reference-named surface (metropolismontecarlo_amd/api.py): the body of
Loop() (Ewald/main.jl:487-644) written with the reference's own calls and host-array mutations,
compared with the oracle step by step."""
import numpy as np
import pytest

import common
from common import rel
from metropolismontecarlo_amd import structs
from metropolismontecarlo_amd.api import (CoulombReal, EwaldReal, EwaldSelf, EwaldShort,
                                          LJ_poly_ΔU, PrepareEwaldVariables, RecipLong, RecipMove,
                                          potential, release_sessions)
from metropolismontecarlo_amd.structs import EWALD, Properties, Properties2, Requirements, Tables

pytestmark = pytest.mark.gpu
TOL = 1e-9
RCUT = 10.0


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(autouse=True)
def _cleanup():
    yield
    release_sessions()


def reference_setup(a):
    """What Ewald/main.jl:242-303 builds: moa/soa, vdwTable, a dummy EWALD + PrepareEwaldVariables,
    totProps."""
    from metropolismontecarlo_amd import io as mio
    moa = structs.make_moa(a["com"].copy(), a["first_atom"], a["last_atom"])
    soa = structs.make_soa(a["coords"].copy(), a["atype"], a["charge"])
    vdwTable = Tables([mio.SPCE_EPS_O, 0.0], [mio.SPCE_SIGMA_O, 0.0])
    box = a["box"]
    ewald = EWALD(5.6 / box, 5, 27, 1, [[1, 1, 1]] * 3, [0.0, 0.0], np.zeros(2, complex),
                  np.zeros(2, complex), structs.factor)                  # main.jl:290-301
    ewald = PrepareEwaldVariables(ewald, box)                            # main.jl:303
    totProps = Properties2(298.15, 0.0331, 0.0, 0.3166, 0.05, 0.3, 0, 0, [], RCUT, RCUT, box)
    return moa, soa, vdwTable, ewald, totProps, box


@pytest.mark.parametrize("k,variant", [(1, "reference"), (4, "unwrapped")])
def test_loop_body_with_reference_calls(k, variant, orc):
    a = common.nist_arrays(k, variant)
    g = common.golden(k, variant)
    moa, soa, vdwTable, ewald, totProps, box = reference_setup(a)
    assert ewald.NKVECS == 337 and ewald.kxyz.shape == (337, 3)

    total = potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")  # main.jl:408
    assert rel(total.energy, g["totals_ewald"]["energy"]) < TOL
    assert rel(total.virial, g["totals_ewald"]["virial"]) < TOL
    assert rel(total.coulomb, g["totals_ewald"]["coulomb"]) < TOL
    assert np.array_equal(ewald.sumQExpOld, ewald.sumQExpNew) and ewald.sumQExpOld.any()
    running = total.energy

    for mv in g["moves"]:
        i = mv["mol"]
        f, l = moa.firstAtom[i - 1], moa.lastAtom[i - 1]
        partial_old_e, partial_old_v = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)   # :491
        e, v, overlap1 = EwaldShort(i, moa, soa, totProps, ewald, box)               # :501
        partial_old_v += v
        partial_old_e += e
        rm_old = moa.COM[i - 1].copy()                                               # :514
        ra_old = soa.coords[f - 1:l].copy()                                          # :515
        moa.COM[i - 1] = mv["com_new"]                                               # :527
        soa.coords[f - 1:l] = mv["atoms_new"]                                        # :552
        ra_new = soa.coords[f - 1:l].copy()
        partial_new_e, partial_new_v = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)   # :557
        e, v, overlap2 = EwaldShort(i, moa, soa, totProps, ewald, box)               # :566
        partial_new_v += v
        partial_new_e += e
        overlap = overlap1 or overlap2
        if not overlap:
            deltaRecip, ewald = RecipMove(box, ewald, ra_old, ra_new, soa.charge[f - 1:l])  # :581
        else:
            deltaRecip = 0.0
        delta = partial_new_e - partial_old_e + deltaRecip                           # :593
        d_ref = mv["d"]
        assert int(overlap) == mv["overlap"]
        assert abs(delta - (d_ref[0] + d_ref[1] + d_ref[2])) < TOL * 1e5
        assert abs((partial_new_v - partial_old_v) + deltaRecip / 3 - d_ref[3]) < TOL * 1e6
        if mv["accept"]:
            running += delta
            ewald.sumQExpOld = [item for item in ewald.sumQExpNew]                   # :621
            ewald.sumQExpOld = np.array(ewald.sumQExpOld)
        else:
            moa.COM[i - 1] = rm_old                                                  # :623
            soa.coords[f - 1:l] = ra_old                                             # :624
            ewald.sumQExpNew = np.array([item for item in ewald.sumQExpOld])         # :628
        assert rel(np.abs(ewald.sumQExpOld).sum(), mv["sum_abs_S_old"]) < 1e-11

    # Poly/main.jl:232-235 invariant through the reference surface
    total2 = potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")
    assert rel(running, total2.energy) < TOL
    s = common.oracle_system(dict(a, com=moa.COM, coords=soa.coords))
    to = orc.potential_ewald(s, orc.Ewald(5.6 / box, 5, 27, box), RCUT, RCUT)
    assert rel(total2.energy, to["energy"]) < TOL


def test_individual_reference_signatures(orc):
    a = common.nist_arrays(2, "reference")
    moa, soa, vdwTable, ewald, totProps, box = reference_setup(a)
    s = common.oracle_system(a)
    ewo = orc.Ewald(5.6 / box, 5, 27, box)
    # EwaldReal moa/soa form
    e, ov = EwaldReal(7, moa, soa, ewald, RCUT, box)
    eo, ovo = orc.ewald_real(7, s, ewo.kappa, RCUT)
    assert ov == ovo and rel(e, eo) < TOL
    # RecipLong(ewald, r, qq_q, box) -> (energy, ewald), fills both arrays
    energy, ew2 = RecipLong(ewald, soa.coords, soa.charge, box)
    assert ew2 is ewald and rel(energy, orc.recip_long(ewo, s.coords, s.charge, box)) < TOL
    assert np.abs(ewald.sumQExpNew - ewo.sumQExpNew).max() < 1e-11 * np.abs(ewo.sumQExpNew).max()
    assert rel(EwaldSelf(ewald, soa.charge), orc.ewald_self(ewo, s.charge)) < 1e-13
    # Wolf total: the 6-argument potential (energy.jl:864-943)
    w = potential(moa, soa, Properties(), ewald, vdwTable, totProps)
    wo = orc.potential_wolf(s, ewo, RCUT, RCUT, literal_prefactor=False)
    assert rel(w.energy, wo["energy"]) < TOL and rel(w.coulomb, wo["coulomb"]) < TOL
    # legacy Requirements forms (energy.jl:126-206, ewalds.jl:205-289, energy.jl:618-711)
    tma = np.stack([a["first_atom"], a["last_atom"]], axis=1)
    system = Requirements(a["com"].copy(), a["coords"].copy(), len(a["com"]), len(a["coords"]),
                          len(a["coords"]), tma, [], [], [], a["atype"], vdwTable, box, RCUT)
    p, v = LJ_poly_ΔU(5, system)
    po, vo = orc.lj_poly_du(5, s, RCUT)
    assert rel(p, po) < TOL and rel(v, vo, abs(po)) < TOL
    e, ov = EwaldReal(system.ra, a["charge"], ewald.kappa, box, tma, 5, system)
    eo, ovo = orc.ewald_real(5, s, ewo.kappa, RCUT, ovr=1.0)
    assert ov == ovo and rel(e, eo) < TOL
    e, ov = CoulombReal(system.ra, a["charge"], box, 5, system)
    eo, ovo = orc.coulomb_real(5, s, RCUT)
    assert ov == ovo and rel(e, eo) < TOL
    energy, _ = RecipLong(system, ewald, system.ra, a["charge"])
    assert rel(energy, orc.recip_long(ewo, s.coords, s.charge, box)) < TOL


def test_reference_asserts_surface_as_assertion_errors():
    a = common.nist_arrays(1)
    moa, soa, vdwTable, ewald, totProps, box = reference_setup(a)
    bad = EWALD(0.28, 5, 26, 1, [[1, 1, 1]], [0.0], [0j], [0j], structs.factor)
    with pytest.raises(AssertionError):
        PrepareEwaldVariables(bad, box)                       # ewalds.jl:49
    with pytest.raises(AssertionError):
        RecipMove(box, ewald, np.zeros((2, 3)), np.zeros((2, 3)), np.zeros(2))   # ewalds.jl:740
    ew4 = PrepareEwaldVariables(EWALD(0.28, 4, 27, 1, [[1, 1, 1]], [0.0], [0j], [0j],
                                      structs.factor), box)
    with pytest.raises(AssertionError):
        RecipMove(box, ew4, np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3))     # nk == 5, :743


# ---- the engine behind the call surface: cached evaluation, stale molecules, named S buffers --------
def _ctx_stats(ewald):
    import ctypes
    s = ewald._session
    st = (ctypes.c_int64 * 10)()
    assert s._L.mmc_ctx_stats(s._h, st) == 0
    return dict(zip(("cmds", "launches", "retries", "cache_hits", "spec_hits", "spec_miss",
                     "launch_evals", "alive", "look_ahead_hits", "look_ahead_posted"), list(st)))


def _oracle_state(a, moa, soa, orc):
    s = common.oracle_system(dict(a, com=moa.COM, coords=soa.coords))
    return s


def test_cache_invalidates_when_loop_restores_a_rejected_molecule(orc):
    """Loop() restores COM/atoms of a rejected molecule (main.jl:623-624): the evaluation cached
    for the moved state must not answer the next call, and the second of LJ_poly_dU(i) /
    EwaldShort(i) on an unchanged system must not reach the device."""
    a = common.nist_arrays(4, "unwrapped")
    moa, soa, vdwTable, ewald, totProps, box = reference_setup(a)
    potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")
    ewo = orc.Ewald(5.6 / box, 5, 27, box)
    rng = np.random.default_rng(3)
    for step, i in enumerate((11, 12, 12, 400, 11)):
        f, l = moa.firstAtom[i - 1], moa.lastAtom[i - 1]
        s0 = _oracle_state(a, moa, soa, orc)
        st0 = _ctx_stats(ewald)
        e_old, v_old = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
        q_old, w_old, _ = EwaldShort(i, moa, soa, totProps, ewald, box)
        st1 = _ctx_stats(ewald)
        assert st1["cmds"] + st1["launch_evals"] - st0["cmds"] - st0["launch_evals"] <= 1
        # EwaldShort came from the cache (and LJ_poly_dU too when nothing changed since molecule
        # i was last evaluated: step 2 repeats the molecule of an accepted step 1)
        assert st1["cache_hits"] == st0["cache_hits"] + (2 if step == 2 else 1)
        eo, vo = orc.lj_poly_du(i, s0, RCUT)
        qo, wo, _ = orc.ewald_short(i, s0, ewo, RCUT)
        assert rel(e_old, eo) < TOL and rel(q_old, qo) < TOL
        rm_old, ra_old = moa.COM[i - 1].copy(), soa.coords[f - 1:l].copy()
        d = (rng.random(3) - 0.5) * 0.3
        moa.COM[i - 1] += d
        soa.coords[f - 1:l] += d
        s1 = _oracle_state(a, moa, soa, orc)
        e_new, _ = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
        q_new, _, _ = EwaldShort(i, moa, soa, totProps, ewald, box)
        en, _ = orc.lj_poly_du(i, s1, RCUT)
        qn, _, _ = orc.ewald_short(i, s1, ewo, RCUT)
        assert rel(e_new, en) < TOL and rel(q_new, qn) < TOL and e_new != e_old
        if step % 2 == 0:   # rejected: Loop restores the molecule (main.jl:623-624)
            moa.COM[i - 1] = rm_old
            soa.coords[f - 1:l] = ra_old
            e_again, _ = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
            q_again, _, _ = EwaldShort(i, moa, soa, totProps, ewald, box)
            assert rel(e_again, eo) < TOL and rel(q_again, qo) < TOL
            assert e_again != e_new                            # not the cached moved state
    assert _ctx_stats(ewald)["retries"] == 0


def test_recip_move_array_roles_follow_the_callers_arrays(orc):
    """RecipMove twice without a commit (sumQExpNew accumulates in place, ewalds.jl:805-814), a
    commit, a rollback, and arrays the context has never seen -- each against the oracle doing
    the same to its own EWALD."""
    a = common.nist_arrays(1, "reference")
    moa, soa, vdwTable, ewald, totProps, box = reference_setup(a)
    potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")
    s = common.oracle_system(a)
    ewo = orc.Ewald(5.6 / box, 5, 27, box)
    orc.recip_long(ewo, s.coords, s.charge, box)
    q = a["charge"][:3]
    rng = np.random.default_rng(5)

    def both(r_old, r_new):
        de, _ = RecipMove(box, ewald, r_old, r_new, q)
        deo = orc.recip_move(box, ewo, r_old, r_new, q)
        assert abs(de - deo) < TOL * 1e5
        scale = np.abs(ewo.sumQExpNew).max()
        assert np.abs(ewald.sumQExpNew - ewo.sumQExpNew).max() < 1e-11 * scale
        assert np.abs(ewald.sumQExpOld - ewo.sumQExpOld).max() < 1e-11 * scale

    r0 = a["coords"][:3].copy()
    r1 = r0 + (rng.random(3) - 0.5)
    r2 = r1 + (rng.random(3) - 0.5)
    both(r0, r1)
    both(r1, r2)                                  # unsettled: in place on sumQExpNew
    ewald.sumQExpOld = ewald.sumQExpNew.copy()    # main.jl:621
    ewo.sumQExpOld = ewo.sumQExpNew.copy()
    both(r2, r0)
    ewald.sumQExpNew = ewald.sumQExpOld.copy()    # main.jl:628
    ewo.sumQExpNew = ewo.sumQExpOld.copy()
    both(r2, r1)
    # arrays from somewhere else
    ewald.sumQExpOld = ewald.sumQExpOld * 0.5 + 0.25j
    ewald.sumQExpNew = ewald.sumQExpNew * 2.0 - 0.5
    ewo.sumQExpOld = ewald.sumQExpOld.copy()
    ewo.sumQExpNew = ewald.sumQExpNew.copy()
    both(r1, r0)
    both(r0, r2)


def test_many_stale_molecules_and_an_idle_server(orc):
    """More molecules changed between two calls than one command carries, a pause longer than
    the host's re-launch threshold, and one longer than the server's own bounded wait."""
    import time
    a = common.nist_arrays(2, "unwrapped")
    with common.device_context(a) as ctx:
        s = common.oracle_system(a)
        ewo = orc.Ewald(5.6 / a["box"], 5, 27, a["box"])
        rng = np.random.default_rng(9)
        for pause in (0.0, 0.4, 1.3):
            for m in rng.choice(np.arange(1, 201), size=7, replace=False):
                d = (rng.random(3) - 0.5) * 0.2
                s.com[m - 1] += d
                s.coords[3 * (m - 1):3 * m] += d
                ctx.set_molecule(int(m), s.com[m - 1], s.coords[3 * (m - 1):3 * m])
            time.sleep(pause)
            for i in (3, 77, 200):
                p, v = ctx.lj_poly_du(i, RCUT)
                e, w, ov = ctx.ewald_short(i, RCUT)
                po, vo = orc.lj_poly_du(i, s, RCUT)
                eo, wo, ovo = orc.ewald_short(i, s, ewo, RCUT)
                assert rel(p, po) < TOL and rel(v, vo, abs(po)) < TOL and rel(e, eo) < TOL and ov == ovo
        com, coords = ctx.download_system()
        assert np.array_equal(com, s.com) and np.array_equal(coords, s.coords)
        st = ctx.stats()
        assert st["launches"] >= 3 and st["cmds"] > 0


def test_server_and_launch_paths_agree(orc):
    a = common.nist_arrays(3, "unwrapped")
    res = []
    for server in (1, 0):
        with common.device_context(a) as ctx:
            ctx.set_option("server", server)
            ctx.recip_long()
            out = []
            for i in (1, 150, 300):
                out.append(ctx.lj_poly_du(i, RCUT) + ctx.ewald_short(i, RCUT)[:2])
            mv = common.golden(3, "unwrapped")["moves"][0]
            d, ov = ctx.trial_move(mv["mol"], mv["com_new"], mv["atoms_new"], RCUT, RCUT)
            ctx.accept_move() if not ov else ctx.reject_move()
            out.append(tuple(d))
            out.append(ctx.lj_poly_du(mv["mol"], RCUT))
            st = ctx.stats()
            assert (st["cmds"] > 0) == bool(server) and (st["launch_evals"] > 0) != bool(server)
            res.append(np.concatenate([np.ravel(x) for x in out]))
    scale = np.maximum(np.abs(res[0]), 1.0)
    assert (np.abs(res[0] - res[1]) / scale).max() < 1e-11


def test_look_ahead_answers_the_next_molecule_of_a_sweep(orc):
    """Loop() sweeps the molecules in order (main.jl:490).  The command that evaluates a moved
    molecule i also has molecule i + 1 evaluated by the server's second set of workgroups; when
    the move is accepted (nothing changes before the next call) LJ_poly_dU(i + 1) / EwaldShort(i + 1)
    are answered from those records -- the same values a command of their own gives -- and when
    it is rejected (the molecule is restored) they are not used."""
    a = common.nist_arrays(4, "unwrapped")
    moa, soa, vdwTable, ewald, totProps, box = reference_setup(a)
    potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")
    ewo = orc.Ewald(5.6 / box, 5, 27, box)
    rng = np.random.default_rng(12)
    direct = {}
    for i in range(1, 41):
        f, l = moa.firstAtom[i - 1], moa.lastAtom[i - 1]
        s0 = _oracle_state(a, moa, soa, orc)
        st0 = _ctx_stats(ewald)
        e_old, _ = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
        q_old, _, _ = EwaldShort(i, moa, soa, totProps, ewald, box)
        st1 = _ctx_stats(ewald)
        eo, _ = orc.lj_poly_du(i, s0, RCUT)
        qo, _, _ = orc.ewald_short(i, s0, ewo, RCUT)
        assert rel(e_old, eo) < TOL and rel(q_old, qo) < TOL, i
        accepted_before = i > 1 and (i - 1) % 4 != 0
        if i > 1:   # answered by the look-ahead exactly when the previous move was kept
            assert (st1["look_ahead_hits"] - st0["look_ahead_hits"] == 1) == accepted_before, i
            assert (st1["cmds"] - st0["cmds"] == 0) == accepted_before, i
        direct[i] = (e_old, q_old)
        rm_old, ra_old = moa.COM[i - 1].copy(), soa.coords[f - 1:l].copy()
        d = (rng.random(3) - 0.5) * 0.3
        moa.COM[i - 1] += d
        soa.coords[f - 1:l] += d
        LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
        EwaldShort(i, moa, soa, totProps, ewald, box)
        if i % 4 == 0:  # rejected: Loop restores the molecule
            moa.COM[i - 1] = rm_old
            soa.coords[f - 1:l] = ra_old
    st = _ctx_stats(ewald)
    assert st["look_ahead_posted"] == 40 and st["look_ahead_hits"] == 30 and st["retries"] == 0
    # the look-ahead's numbers are the numbers of a command of its own: same kernel, plan and sums
    release_sessions()
    import os
    os.environ["MMC_CTX_LOOKAHEAD"] = "0"
    try:
        moa, soa, vdwTable, ewald, totProps, box = reference_setup(a)
        potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")
        rng = np.random.default_rng(12)
        for i in range(1, 41):
            f, l = moa.firstAtom[i - 1], moa.lastAtom[i - 1]
            e_old, _ = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
            q_old, _, _ = EwaldShort(i, moa, soa, totProps, ewald, box)
            assert (e_old, q_old) == direct[i], i          # bit for bit
            rm_old, ra_old = moa.COM[i - 1].copy(), soa.coords[f - 1:l].copy()
            d = (rng.random(3) - 0.5) * 0.3
            moa.COM[i - 1] += d
            soa.coords[f - 1:l] += d
            LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
            EwaldShort(i, moa, soa, totProps, ewald, box)
            if i % 4 == 0:
                moa.COM[i - 1] = rm_old
                soa.coords[f - 1:l] = ra_old
        assert _ctx_stats(ewald)["look_ahead_posted"] == 0
    finally:
        os.environ.pop("MMC_CTX_LOOKAHEAD", None)
