"""GPU parity for the volume-move path (BASELINE config 4: K6 + K2 + K3 per volume move) and the
large synthetic systems of configs 4 and 5.  The reference's only statement of the move is the
docstring Ewald/volumeChange.jl:8-150; the energies at the new volume are checked against the
oracle evaluated on coordinates rescaled on the host by the same rule."""
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


def host_rescale(a, L_new):
    """volumeChange.jl:62-80: COMs scale by f, atoms translate with their molecule."""
    f = L_new / a["box"]
    com = a["com"] * f
    d = com - a["com"]
    coords = a["coords"] + np.repeat(d, 3, axis=0)
    return dict(a, com=com, coords=coords, box=float(L_new))


def water_lattice(n_mol, geometry="spce", rho=0.033101144):
    from metropolismontecarlo_amd import io as mio, structs
    box, com, coords = mio.cubic_lattice_water(n_mol, rho, geometry, seed=11234)
    if geometry == "spce":
        q = [mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H]
        tab = structs.Tables([mio.SPCE_EPS_O, 0.0], [mio.SPCE_SIGMA_O, 0.0])
    else:  # TIP3P: water.top:13-14,25-27 (sigma 0.315061 nm, eps 0.6364 kJ/mol, q -0.834/+0.417)
        q = [-0.834, 0.417, 0.417]
        tab = structs.Tables([0.6364 / structs.R, 0.0], [3.15061, 0.0])
    first = 3 * np.arange(n_mol, dtype=np.int64) + 1
    return dict(com=com, first_atom=first, last_atom=first + 2, coords=coords,
                atype=np.tile([1, 2, 2], n_mol), charge=np.tile(q, n_mol), eps=tab.eps_ij,
                sig=tab.sig_ij, box=box)


@pytest.mark.parametrize("k,variant,dL", [(1, "unwrapped", 0.5), (4, "unwrapped", -0.3),
                                          (4, "reference", 0.2)])
def test_volume_change_context(k, variant, dL, orc):
    a = common.nist_arrays(k, variant)
    L_new = a["box"] + dL
    a2 = host_rescale(a, L_new)
    s2 = common.oracle_system(a2)
    to = orc.potential_ewald(s2, orc.Ewald(5.6 / L_new, 5, 27, L_new), RCUT, RCUT)
    with common.device_context(a) as ctx:
        ctx.potential_ewald(RCUT, RCUT)
        ctx.volume_change(L_new, 5.6 / L_new)
        com, coords = ctx.download_system()
        assert np.array_equal(com, a2["com"]) and np.array_equal(coords, a2["coords"])
        kxyz, cfac = ctx.get_kvectors()
        ew = orc.Ewald(5.6 / L_new, 5, 27, L_new)
        assert np.array_equal(kxyz, ew.kxyz) and np.allclose(cfac, ew.cfac, rtol=1e-14)
        t = ctx.potential_ewald(RCUT, RCUT)
        for key in ("energy", "virial", "lj", "real", "recip", "self"):
            assert rel(t[key], to[key]) < TOL, key
        # per-move calls keep working in the new box
        p, v = ctx.lj_poly_du(3, RCUT)
        po, vo = orc.lj_poly_du(3, s2, RCUT)
        assert rel(p, po) < TOL
        e, ov = ctx.ewald_real(3, RCUT)
        eo, ovo = orc.ewald_real(3, s2, 5.6 / L_new, RCUT)
        assert ov == ovo and rel(e, eo) < TOL


def test_volume_change_batch_then_moves(orc):
    from test_gpu_batch import make_batch
    a = common.nist_arrays(4, "unwrapped")
    L_new = a["box"] * 1.004
    a2 = host_rescale(a, L_new)
    to = orc.potential_ewald(common.oracle_system(a2), orc.Ewald(5.6 / L_new, 5, 27, L_new),
                             RCUT, RCUT)
    with make_batch(a, 3) as b:
        b.potential_ewald()
        b.volume_change(L_new, 5.6 / L_new)
        t = b.potential_ewald()
        for r in range(3):
            assert rel(t[r]["energy"], to["energy"]) < TOL
            com, coords, _ = b.get_replica(r)
            assert np.array_equal(com, a2["com"]) and np.array_equal(coords, a2["coords"])
        # the driver's host mirror was rescaled with the same arithmetic: running totals stay exact
        e0 = np.array([x["energy"] for x in t])
        e1, st = b.run(300, 298.15, 0.316555789, 0.05, seed=5, energies=e0, n_groups=2,
                       n_threads=2)
        e2 = np.array([x["energy"] for x in b.potential_ewald()])
        assert np.abs(e1 - e2).max() < 1e-9 * np.abs(e2).max()


def test_npt_volume_move_host_logic(orc):
    """metropolismontecarlo_amd.npt.VolumeChange (volumeChange.jl:8-150): a rejected move restores
    coordinates, tables and S(k) exactly; an accepted one leaves the rescaled system."""
    from metropolismontecarlo_amd.npt import VolumeChange
    a = common.nist_arrays(1, "unwrapped")

    class Fixed:  # scripted "random" numbers
        def __init__(self, vals): self.vals = list(vals)
        def random(self): return self.vals.pop(0)

    with common.device_context(a) as ctx:
        t0 = ctx.potential_ewald(RCUT, RCUT)
        com0, coords0 = ctx.download_system()
        S0 = ctx.get_sumqexp()[0]
        # expansion by +200 A^3 at a huge pressure -> rejected
        acc, box, e, _ = VolumeChange(ctx, t0["energy"], a["box"], 100, 1e6, 298.15, 400.0, RCUT,
                                      RCUT, Fixed([1.0, 0.999999]))
        assert not acc and box == a["box"] and e == t0["energy"]
        com1, coords1 = ctx.download_system()
        assert np.array_equal(com0, com1) and np.array_equal(coords0, coords1)
        assert np.abs(ctx.get_sumqexp()[0] - S0).max() < 1e-12 * np.abs(S0).max()
        t1 = ctx.potential_ewald(RCUT, RCUT)
        assert rel(t1["energy"], t0["energy"]) < 1e-13
        # the same expansion at zero pressure with rand = 0 -> accepted whatever dE is
        acc, box, e, tot = VolumeChange(ctx, t0["energy"], a["box"], 100, 0.0, 298.15, 400.0,
                                        RCUT, RCUT, Fixed([1.0, 0.0]))
        assert acc and box == pytest.approx((a["box"] ** 3 + 200.0) ** (1 / 3), rel=1e-15)
        a2 = host_rescale(a, box)
        to = orc.potential_ewald(common.oracle_system(a2), orc.Ewald(5.6 / box, 5, 27, box),
                                 RCUT, RCUT)
        assert rel(e, to["energy"]) < TOL and rel(tot["recip"], to["recip"]) < TOL
        # a contraction below 2 r_cut is refused before touching the device
        acc, box2, _, _ = VolumeChange(ctx, e, box, 100, 0.0, 298.15, 4000.0, RCUT, RCUT,
                                       Fixed([0.0, 0.0]))
        assert not acc and box2 == box


@pytest.mark.parametrize("n_mol,geometry", [(10000, "spce"), (5000, "tip3p")])
def test_large_synthetic_boxes(n_mol, geometry, orc):
    """BASELINE config 4 (10 000 SPC/E, volume perturbation +0.5 % -> K6+K2+K3) and config 5
    (5 000 TIP3P, Ewald vs the reference's Wolf total) against the oracle."""
    a = water_lattice(n_mol, geometry)
    s = common.oracle_system(a)
    box = a["box"]
    with common.device_context(a) as ctx:
        t = ctx.potential_ewald(RCUT, RCUT)
        to = orc.potential_ewald(s, orc.Ewald(5.6 / box, 5, 27, box), RCUT, RCUT)
        for key in ("energy", "virial", "lj", "real", "recip", "self"):
            assert rel(t[key], to[key], 1.0) < TOL, key
        assert t["n_overlap"] == to["n_overlap"]
        if geometry == "tip3p":
            w = ctx.potential_wolf(RCUT, RCUT)
            wo = orc.potential_wolf(s, orc.Ewald(5.6 / box, 5, 27, box), RCUT, RCUT,
                                    literal_prefactor=False)
            for key in ("energy", "coulomb", "real", "self"):
                assert rel(w[key], wo[key], 1.0) < TOL, key
        else:
            L_new = (1.005 * box ** 3) ** (1 / 3)
            a2 = host_rescale(a, L_new)
            ctx.volume_change(L_new, 5.6 / L_new)
            t2 = ctx.potential_ewald(RCUT, RCUT)
            to2 = orc.potential_ewald(common.oracle_system(a2),
                                      orc.Ewald(5.6 / L_new, 5, 27, L_new), RCUT, RCUT)
            for key in ("energy", "lj", "real", "recip", "self"):
                assert rel(t2[key], to2[key], 1.0) < TOL, key


def test_npt_accept_reject_cycle_at_10000_molecules(orc):
    """A full volume move both ways at BASELINE configs[3]'s size without a host round trip
    (mmc_volume_trial / accept / reject).  The rejected move must give back coordinates,
    structure factors and everything the per-molecule kernels read (records, fixed-point centres
    of mass, erfc table: seen through LJ_poly_dU / EwaldShort on the context's server) BIT FOR
    BIT; the accepted one leaves the rescaled system, checked against the oracle."""
    from metropolismontecarlo_amd.npt import VolumeChange
    a = water_lattice(10000, "spce")
    box = a["box"]

    class Fixed:
        def __init__(self, vals): self.vals = list(vals)
        def random(self): return self.vals.pop(0)

    with common.device_context(a) as ctx:
        t0 = ctx.potential_ewald(RCUT, RCUT)
        com0, coords0 = ctx.download_system()
        S0 = ctx.get_sumqexp()[0]
        probe0 = [ctx.lj_poly_du(i, RCUT) + ctx.ewald_short(i, RCUT) for i in (1, 5000, 10000)]
        vmax = 0.02 * box ** 3                       # +1 % in volume with rand = 1
        acc, b1, e1, _ = VolumeChange(ctx, t0["energy"], box, 10000, 1e9, 298.15, vmax, RCUT, RCUT,
                                      Fixed([1.0, 0.999999]))
        assert not acc and b1 == box and e1 == t0["energy"]
        com1, coords1 = ctx.download_system()
        assert np.array_equal(com0, com1) and np.array_equal(coords0, coords1)
        assert np.array_equal(ctx.get_sumqexp()[0], S0)
        assert [ctx.lj_poly_du(i, RCUT) + ctx.ewald_short(i, RCUT) for i in (1, 5000, 10000)] == probe0
        t1 = ctx.potential_ewald(RCUT, RCUT)
        assert all(t1[k] == t0[k] for k in ("energy", "virial", "lj", "real", "recip", "self"))
        kx, cf = ctx.get_kvectors()
        ew0 = orc.Ewald(5.6 / box, 5, 27, box)
        assert np.array_equal(kx, ew0.kxyz) and np.allclose(cf, ew0.cfac, rtol=1e-14)
        # the same move accepted (zero pressure, rand = 0)
        acc, b2, e2, tot = VolumeChange(ctx, t0["energy"], box, 10000, 0.0, 298.15, vmax, RCUT, RCUT,
                                        Fixed([1.0, 0.0]))
        assert acc and b2 == pytest.approx((1.01 * box ** 3) ** (1 / 3), rel=1e-14)
        a2 = host_rescale(a, b2)
        to = orc.potential_ewald(common.oracle_system(a2), orc.Ewald(5.6 / b2, 5, 27, b2), RCUT, RCUT)
        for key in ("energy", "lj", "real", "recip", "self"):
            assert rel(tot[key], to[key], 1.0) < TOL, key
        com2, coords2 = ctx.download_system()
        assert np.array_equal(com2, a2["com"]) and np.array_equal(coords2, a2["coords"])
        # per-molecule calls in the new box: the mirror of the context followed the device
        s2 = common.oracle_system(a2)
        p, _ = ctx.lj_poly_du(77, RCUT)
        po, _ = orc.lj_poly_du(77, s2, RCUT)
        assert rel(p, po) < TOL
        ctx.set_molecule(77, a2["com"][76] + 0.05, a2["coords"][228:231] + 0.05)
        s2.com[76] += 0.05
        s2.coords[228:231] += 0.05
        e, _, ov = ctx.ewald_short(77, RCUT)
        eo, _, ovo = orc.ewald_short(77, s2, orc.Ewald(5.6 / b2, 5, 27, b2), RCUT)
        assert ov == ovo and rel(e, eo) < TOL


def test_context_server_of_a_large_system_equals_launch_per_evaluation():
    """10 000 molecules: the context server takes 21 workgroups (plus 21 looking ahead); the same
    trial moves with a launch per evaluation (option "server" = 0) give the same terms to 1e-12,
    accepted and rejected moves mixed, and the structure factors agree at the end."""
    a = water_lattice(10000, "spce")
    rng_moves = np.random.default_rng(3).random((60, 3))
    res = {}
    for server in (1, 0):
        with common.device_context(a) as ctx:
            ctx.set_option("server", -1 if server else 0)
            c, x = a["com"].copy(), a["coords"].copy()
            out = []
            for k in range(60):
                i = (k * 167) % 10000 + 1
                d = (rng_moves[k] - 0.5) * 0.3
                cn, an = c[i - 1] + d, x[3 * (i - 1):3 * i] + d
                du, ov = ctx.trial_move(i, cn, an, RCUT, RCUT)
                out.append(du.copy())
                if k % 3 and not ov:
                    ctx.accept_move()
                    c[i - 1] = cn
                    x[3 * (i - 1):3 * i] = an
                else:
                    ctx.reject_move()
            st = ctx.stats()
            assert (st["cmds"] > 0) == bool(server)
            res[server] = (np.array(out), ctx.get_sumqexp()[0].copy())
    scale = np.abs(res[0][0]).max() + 1e4
    assert np.abs(res[1][0] - res[0][0]).max() < 1e-12 * scale
    assert np.abs(res[1][1] - res[0][1]).max() < 1e-11 * np.abs(res[0][1]).max()


def test_nothing_else_runs_between_a_volume_trial_and_its_decision():
    """Between mmc_volume_trial and accept / reject the context holds the trial volume: per-molecule
    calls, updates and trial moves are refused (they would pull the host mirror to the trial volume,
    which a rejection cannot undo); after the rejection everything answers as before the trial."""
    from metropolismontecarlo_amd._lib import MMCError
    a = common.nist_arrays(1, "unwrapped")
    box = a["box"]
    with common.device_context(a) as ctx:
        ctx.potential_ewald(RCUT, RCUT)
        before = ctx.lj_poly_du(3, RCUT) + ctx.ewald_short(3, RCUT)
        L1 = box + 0.4
        ctx.volume_trial(L1, 5.6 / L1, RCUT, RCUT)
        for call in (lambda: ctx.lj_poly_du(3, RCUT), lambda: ctx.ewald_short(3, RCUT),
                     lambda: ctx.set_molecule(3, a["com"][2], a["coords"][6:9]),
                     lambda: ctx.update_system(a["com"], a["coords"]),
                     lambda: ctx.trial_move(3, a["com"][2], a["coords"][6:9], RCUT, RCUT),
                     lambda: ctx.recip_long(),
                     lambda: ctx.volume_trial(L1, 5.6 / L1, RCUT, RCUT)):
            with pytest.raises(MMCError, match="MMC_ERR_STATE"):
                call()
        ctx.volume_reject()
        assert ctx.lj_poly_du(3, RCUT) + ctx.ewald_short(3, RCUT) == before
        com, coords = ctx.download_system()
        assert np.array_equal(com, a["com"]) and np.array_equal(coords, a["coords"])


# ---- NPT on the batch: trial moves and volume moves on ONE device state (BASELINE configs[3]) ------
def make_one_replica_batch(a, rcut=RCUT):
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Batch
    b = Batch(1, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
              5.6 / a["box"], structs.factor, rcut, rcut)
    b.set_option("device_moves", 1)
    return b


def test_batch_volume_move_at_10000_molecules_reject_restores_accept_matches_the_oracle(orc):
    """mmc_batch_volume_trial / reject / accept on the one-replica batch of BASELINE configs[3], with
    trial moves on the move server before and after: a rejected volume move gives back coordinates,
    S(k) and the total energy BIT FOR BIT and the chain goes on as if it had not happened; an
    accepted one leaves the rescaled system (oracle) and the chain goes on in the new box."""
    from metropolismontecarlo_amd._lib import MMCError
    a = water_lattice(10000, "spce")
    box = a["box"]
    T, dr, dphi = 298.15, 0.316555789, 0.05
    with make_one_replica_batch(a) as b, make_one_replica_batch(a) as ref:
        e0 = b.potential_ewald(as_array=True)["energy"].copy()
        e1, st = b.run(150, T, dr, dphi, 5, e0, n_groups=1)
        assert st["server_steps"] == 150
        r0 = ref.potential_ewald(as_array=True)["energy"].copy()
        r1, _ = ref.run(150, T, dr, dphi, 5, r0, n_groups=1)
        assert np.array_equal(e1, r1)
        t_before = b.potential_ewald(as_array=True).copy()    # (RecipLong: S(k) recomputed from scratch)
        ref.potential_ewald()                                 # ... so the twin does the same
        before = b.get_replica(0)
        L1 = (1.004 * box ** 3) ** (1 / 3)
        tot = b.volume_trial(L1, 5.6 / L1)
        for call in (lambda: b.run(3, T, dr, dphi, 6, e1, n_groups=1),
                     lambda: b.set_replica(0, a["com"], a["coords"]),
                     lambda: b.volume_trial(L1, 5.6 / L1), lambda: b.volume_change(L1, 5.6 / L1)):
            with pytest.raises(MMCError, match="MMC_ERR_STATE"):
                call()
        b.volume_reject()
        after = b.get_replica(0)
        assert all(np.array_equal(x, y) for x, y in zip(before, after))
        t_after = b.potential_ewald(as_array=True)
        ref.potential_ewald()
        assert all(t_after[k][0] == t_before[k][0] for k in ("energy", "lj", "real", "recip", "self"))
        # the chain continues exactly like one that never tried the volume move
        e2, _ = b.run(120, T, dr, dphi, 7, e1, n_groups=1)
        r2, _ = ref.run(120, T, dr, dphi, 7, r1, n_groups=1)
        assert np.array_equal(e2, r2)
        assert all(np.array_equal(x, y) for x, y in zip(b.get_replica(0), ref.get_replica(0)))
        # accepted: the rescaled system, checked against the oracle; then moves in the new box
        com, coords, _ = b.get_replica(0)
        cur = dict(a, com=com, coords=coords)
        tot = b.volume_trial(L1, 5.6 / L1)
        b.volume_accept()
        a2 = host_rescale(cur, L1)
        to = orc.potential_ewald(common.oracle_system(a2), orc.Ewald(5.6 / L1, 5, 27, L1), RCUT, RCUT)
        for key in ("energy", "lj", "real", "recip", "self"):
            assert rel(tot[key], to[key], 1.0) < TOL, key
        com2, coords2, _ = b.get_replica(0)
        assert np.array_equal(com2, a2["com"]) and np.array_equal(coords2, a2["coords"])
        e3, st3 = b.run(100, T, dr, dphi, 8, np.array([tot["energy"]]), n_groups=1)
        assert st3["server_steps"] == 100
        t3 = b.potential_ewald(as_array=True)
        assert rel(e3[0], t3["energy"][0]) < TOL
        com3, coords3, _ = b.get_replica(0)
        to3 = orc.potential_ewald(common.oracle_system(dict(a2, com=com3, coords=coords3)),
                                  orc.Ewald(5.6 / L1, 5, 27, L1), RCUT, RCUT)
        assert rel(t3["energy"][0], to3["energy"]) < TOL
    with make_one_replica_batch(a) as b:
        pass
    # more than one replica: a batch has one box
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Batch
    a1 = common.nist_arrays(1, "unwrapped")
    with Batch(2, a1["com"], a1["coords"], a1["atype"], a1["charge"], a1["eps"], a1["sig"], a1["box"],
               5.6 / a1["box"], structs.factor, 9.0, 9.0) as b2:
        b2.potential_ewald()
        with pytest.raises(MMCError, match="MMC_ERR_UNSUPPORTED"):
            b2.volume_trial(a1["box"] + 0.1, 5.6 / (a1["box"] + 0.1))


def test_npt_chain_on_the_batch_stepped_by_the_oracle(orc):
    """mmc_batch_run_npt on NIST configuration 2 (200 molecules, r_cut 9 A): sweeps of trial moves
    interleaved with volume moves, the whole chain replayed by the oracle -- every proposal rebuilt
    from the Philox draws, dU from orc.trial_move, the volume move's two uniforms from slot
    MMC_SLOT_VOLUME, its energy from orc.potential_ewald on coordinates rescaled on the host
    (volumeChange.jl:59-147) -- must end in the same box, the same coordinates, the same accept
    counts and the same running energy."""
    import math
    from test_gpu_batch import _rigid_proposal
    from test_gpu_moves import philox_pair
    a = common.nist_arrays(2, "unwrapped")
    n_mol, rc = a["com"].shape[0], 9.0
    T, dr, dphi, seed, rep0 = 298.15, 0.316555789, 0.05, 31337, 2
    P, n_sweeps, per_sweep = 0.03, 6, 45
    vmax = 0.03 * a["box"] ** 3
    with make_one_replica_batch(a, rc) as b:
        e0 = float(b.potential_ewald(as_array=True)["energy"][0])
        e1, st, ns = b.run_npt(n_sweeps, T, P, vmax, dr, dphi, seed, e0, moves_per_sweep=per_sweep,
                               replica0=rep0)
        com, coords, S = b.get_replica(0)
        t_end = b.potential_ewald(as_array=True)
    assert st["moves"] == n_sweeps * per_sweep and ns["vol_attempt"] == n_sweeps
    # the oracle's chain
    cur = dict(a)
    s = common.oracle_system(cur)
    box = a["box"]
    ew = orc.Ewald(5.6 / box, 5, 27, box)
    energy = orc.potential_ewald(s, ew, rc, rc)["energy"]
    assert rel(e0, energy) < TOL
    step, n_acc_vol, n_acc = 0, 0, 0
    for sweep in range(n_sweeps):
        for k in range(per_sweep):
            i = k % n_mol                                        # every run restarts its sweep (main.jl:490)
            kind, c_new, a_new, u = _rigid_proposal(seed, rep0, step, s.com[i].copy(),
                                                    s.coords[3 * i:3 * i + 3].copy(), box, dr, dphi)
            d, ov = orc.trial_move(i + 1, s, ew, rc, rc, c_new, a_new)
            delta = d[0] + d[1] + d[2]
            x = delta / T
            if (x < 0.0 or math.exp(-x) > u) and not ov:
                energy += delta
                s.com[i] = c_new
                s.coords[3 * i:3 * i + 3] = a_new
                ew.sumQExpOld = ew.sumQExpNew.copy()
                n_acc += 1
            else:
                ew.sumQExpNew = ew.sumQExpOld.copy()
            step += 1
        ua, ub = philox_pair(seed, rep0, step, 0x40000000)        # MMC_SLOT_VOLUME at the step count reached
        vol_old = box ** 3
        vol_new = vol_old + (ua - 0.5) * vmax                     # volumeChange.jl:59
        L_new = vol_new ** (1.0 / 3.0)
        if rc > L_new / 2:
            continue
        a2 = host_rescale(dict(cur, com=s.com.copy(), coords=s.coords.copy(), box=box), L_new)
        s2 = common.oracle_system(a2)
        ew2 = orc.Ewald(5.6 / L_new, 5, 27, L_new)
        e_new = orc.potential_ewald(s2, ew2, rc, rc)["energy"]
        arg = -(1.0 / T) * (P * (vol_new - vol_old) - n_mol * math.log(vol_new / vol_old) * T
                            + (e_new - energy))                   # :129-130
        if ub < math.exp(min(arg, 700.0)):                        # :132
            s, ew, box, energy, cur = s2, ew2, L_new, e_new, a2
            n_acc_vol += 1
    assert 0 < n_acc_vol < n_sweeps, "pick parameters that accept some volume moves and reject some"
    assert ns["vol_accept"] == n_acc_vol and st["trans_accept"] + st["rot_accept"] == n_acc
    assert ns["box"] == pytest.approx(box, rel=1e-15)
    assert np.abs(com - s.com).max() < 1e-11 and np.abs(coords - s.coords).max() < 1e-11
    assert abs(e1 - energy) < TOL * abs(energy)
    assert rel(t_end["energy"][0], orc.potential_ewald(s, orc.Ewald(5.6 / box, 5, 27, box), rc, rc)["energy"]) < TOL
