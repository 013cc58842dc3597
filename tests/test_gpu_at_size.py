"""Parity at BASELINE.json's own sizes, against the ORACLE (not against another form of the product):

* configs[3] -- 10 000 SPC/E molecules: the native driver's chain on the default move-server shape
  (21 workgroups / 84 parts of k_move_server_lat), on k_move_eval_lat launched per step and on
  k_move_eval_wave, every step's dU and decision checked by the oracle stepping the same chain
  (Ewald/main.jl:487-644, ewalds.jl:718-826, energy.jl:209-290); and the reference's own call
  surface on the context server at that size (LJ_poly_dU / EwaldShort / RecipMove per call).
* configs[4] -- 5 000 TIP3P molecules, Wolf vs Ewald, fp32 vs fp64: tests/cfg5_study.py, the same
  code scripts/precision_study.py runs to write profiles/roundN_cfg5_precision_study.json.
"""
import json
import math
import os

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


@pytest.fixture(scope="module")
def spce_10000():
    from test_gpu_npt import water_lattice
    return water_lattice(10000, "spce")


@pytest.mark.parametrize("kernel,persistent,parts", [(3, 1, 0), (4, 0, 84), (2, 0, 1), (2, 0, 5)])
def test_10000_molecule_chain_stepped_by_the_oracle(kernel, persistent, parts, spce_10000, orc):
    """mmc_batch_run at 10 000 molecules with device-side proposals.  (3, 1, 0) is what a caller
    gets without asking for anything: the latency server in its large-system shape."""
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Batch
    from test_gpu_batch import _rigid_proposal
    a = spce_10000
    n_mol, box = 10000, a["box"]
    n_steps, seed, T, dr, dphi, rep0 = 64, 777, 298.15, 0.316555789, 0.05, 3
    with Batch(1, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], box,
               5.6 / box, structs.factor, RCUT, RCUT) as b:
        b.set_option("device_moves", 1)
        b.set_option("kernel", kernel)
        b.set_option("persistent", persistent)
        b.set_option("trace_steps", n_steps)
        e0 = b.potential_ewald(as_array=True)["energy"].copy()
        e1, st = b.run(n_steps, T, dr, dphi, seed=seed, energies=e0, n_groups=1, n_parts=parts,
                       n_threads=1, replica0=rep0)
        assert st["server_steps"] == (n_steps if persistent else 0) and st["torn_records"] == 0
        d_gpu, f_gpu = b.get_trace(n_steps)
        com, coords, S = b.get_replica(0)
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / box, 5, 27, box)
    orc.recip_long(ew, s.coords, s.charge, box)
    running, n_rej, n_rot = 0.0, 0, 0
    for step in range(n_steps):
        i = step % n_mol                                                  # main.jl:490
        kind, c_new, a_new, u = _rigid_proposal(seed, rep0, step, s.com[i].copy(),
                                                s.coords[3 * i:3 * i + 3].copy(), box, dr, dphi)
        d, ov = orc.trial_move(i + 1, s, ew, RCUT, RCUT, c_new, a_new)
        delta = d[0] + d[1] + d[2]                                        # main.jl:593
        x = delta / T
        accept = (x < 0.0 or math.exp(-x) > u) and not ov                 # main.jl:598
        assert abs(d_gpu[0, step] - delta) < TOL * (abs(delta) + 1e4), (step, kind)
        assert f_gpu[0, step] == (int(accept) | (int(ov) << 1) | (kind << 2)), step
        n_rej += not accept
        n_rot += kind
        if accept:
            running += delta
            s.com[i] = c_new
            s.coords[3 * i:3 * i + 3] = a_new
            ew.sumQExpOld = ew.sumQExpNew.copy()                          # main.jl:621
        else:
            ew.sumQExpNew = ew.sumQExpOld.copy()                          # main.jl:628
    assert n_rej > 3 and n_rot > 10 and n_steps - n_rej > 10
    assert np.abs(com - s.com).max() < 2e-13 and np.abs(coords - s.coords).max() < 2e-13
    assert np.abs(S - ew.sumQExpOld).max() < 1e-11 * np.abs(ew.sumQExpOld).max()
    assert abs((e1[0] - e0[0]) - running) < TOL * 1e5


@pytest.mark.parametrize("server", [-1, 0])
def test_10000_molecule_call_surface_against_the_oracle(server, spce_10000, orc):
    """Loop()'s five calls per trial move (main.jl:491-587) on the context at 10 000 molecules --
    served by the context's persistent kernel (21 + 21 workgroups) or by a launch per evaluation
    -- each call's return value against the oracle's, accepted and rejected moves mixed, sweep
    order (so the look-ahead answers some of the calls) and far-apart molecules."""
    a = spce_10000
    box = a["box"]
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / box, 5, 27, box)
    orc.recip_long(ew, s.coords, s.charge, box)
    rng = np.random.default_rng(11)
    mols = list(range(1, 25)) + [5000, 5001, 10000, 1, 9999]
    with common.device_context(a) as ctx:
        ctx.set_option("server", server)
        ctx.recip_long()
        q = a["charge"][:3].copy()
        n_acc = 0
        for k, i in enumerate(mols):
            sl = slice(3 * (i - 1), 3 * i)
            for state in ("old", "new"):
                if state == "new":
                    d = (rng.random(3) - 0.5) * 0.3
                    r_old = s.coords[sl].copy()
                    c_old = s.com[i - 1].copy()
                    s.com[i - 1] = c_old + d
                    s.coords[sl] = r_old + d
                    ctx.set_molecule(i, s.com[i - 1], s.coords[sl])       # main.jl:527,552
                p, v = ctx.lj_poly_du(i, RCUT)                            # main.jl:491 / :557
                e, ev, ovl = ctx.ewald_short(i, RCUT)                     # main.jl:501 / :566
                po, vo = orc.lj_poly_du(i, s, RCUT)
                eo, evo, ovo = orc.ewald_short(i, s, ew, RCUT)
                assert rel(p, po, 1.0) < TOL and rel(v, vo, 1.0) < TOL, (k, i, state)
                assert ovl == ovo and rel(e, eo, 1.0) < TOL and rel(ev, evo, 1.0) < TOL, (k, i)
            dr = ctx.recip_move(r_old, s.coords[sl], q)                   # main.jl:581
            dro = orc.recip_move(box, ew, r_old, s.coords[sl], q)
            assert abs(dr - dro) < TOL * 1e4, (k, i)
            if k % 3:                                                     # accept (main.jl:621)
                ctx.recip_commit()
                ew.sumQExpOld = ew.sumQExpNew.copy()
                n_acc += 1
            else:                                                         # reject (:623-628)
                s.com[i - 1] = c_old
                s.coords[sl] = r_old
                ctx.set_molecule(i, c_old, r_old)
                ctx.recip_rollback()
                ew.sumQExpNew = ew.sumQExpOld.copy()
        S_old, S_new = ctx.get_sumqexp()
        assert np.abs(S_old - ew.sumQExpOld).max() < 1e-11 * np.abs(ew.sumQExpOld).max()
        st = ctx.stats()
        assert (st["cmds"] > 0) == (server != 0)
        t = ctx.potential_ewald(RCUT, RCUT)
    to = orc.potential_ewald(s, orc.Ewald(5.6 / box, 5, 27, box), RCUT, RCUT)
    assert rel(t["energy"], to["energy"]) < TOL and n_acc > 10


def test_cfg5_tip3p_5000_precision_study_against_the_oracle(orc):
    """BASELINE configs[4] at its size: the fp64 product path and the fp32 / mixed study kernels on
    5 000 TIP3P molecules, Ewald (energy.jl:946-1032) and the reference's Wolf total
    (energy.jl:864-943), totals and 240 scripted trial moves, all against the oracle's fp64.
    The asserted bands are the measured ones with a factor ~3 of room; the JSON the run produces is
    what profiles/roundN_cfg5_precision_study.json is copied from."""
    import cfg5_study
    out = cfg5_study.run(n_moves=240)
    t, du = out["totals"], out["dU"]
    # fp64 product path vs oracle: the parity gate
    for key in ("lj", "real", "recip", "ewald_self", "ewald_total", "wolf_total", "wolf_const"):
        assert t["fp64_vs_oracle_rel"][key] < TOL, key
    assert du["fp64"]["ewald"]["max_abs_err_K"] < TOL * 1e4
    assert du["fp64"]["overlap_flags_differ"] == 0
    # single precision: bands (see profiles/README.md "cfg5")
    for name in ("fp32", "mixed"):
        r = t[name]["rel_err_vs_oracle"]
        assert r["lj"] < 1e-5 and r["real"] < 1e-4 and r["recip"] < 2e-6
        assert r["ewald_total"] < 1e-7 and r["wolf_total"] < 1e-7
        assert du[name]["ewald"]["max_abs_err_K"] < 0.5
        assert du[name]["wolf"]["max_abs_err_K"] < 0.5
        assert du[name]["ewald"]["max_abs_err_K"] > 1e-4      # it really is single precision
        assert du[name]["overlap_flags_differ"] == 0
    assert abs(t["wolf_vs_ewald_fp64"]["per_molecule_K"]) > 100.0   # the two totals are different things
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(dst, exist_ok=True)
        with open(os.path.join(dst, "cfg5_precision_study_test.json"), "w") as fh:
            json.dump(out, fh, indent=1)
    except OSError:
        pass
