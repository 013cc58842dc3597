"""CPU tests of the oracle (oracle/mmc_oracle.c): pin it on every known answer available.

  * NIST SPC/E reference energies for the four bundled configurations (the reference points at
    them, Ewald/main.jl:231-239; values in BASELINE.md): E_fourier and E_self pin
    RecipLong/EwaldSelf/PrepareEwaldVariables exactly; E_real and E_disp pin the erfc and LJ pair
    arithmetic + minimum image through the reference's atom-cutoff variants.
  * the reference's own analytic tests: test_LJ (Monatomic/mainMonatomic.jl:292-325 ==
    Ewald/tests.jl:127-161), test_two_LJ_triangles (Ewald/tests.jl:8-82), test_COM (:86-102).
  * an independent numpy/scipy statement (oracle/numpy_check.py) of the molecular-cutoff
    functions, which have no stored answer anywhere in the reference.
  * the committed golden file (regression pin).
"""
import numpy as np
import pytest

import common
from common import rel
from oracle import numpy_check as npc
from oracle import oracle as orc

RCUT = 10.0


def test_factor_constant():
    # Ewald/constants.jl:24-28; value quoted in SURVEY.md
    assert orc.factor() == 167100.95663229248
    from metropolismontecarlo_amd import structs
    assert structs.factor == orc.factor()


def test_vector1D_reference_semantics():
    L = 5.0
    assert orc.vector1D(0.0, 4.0, L) == -1.0          # image is closer
    assert orc.vector1D(0.0, 2.0, L) == 2.0
    assert orc.vector1D(4.0, 0.0, L) == 1.0
    assert orc.vector1D(1.0, 1.0, L) == 0.0
    # tie at L/2: the comparison is strict, so c1 < c2 gives d - L
    assert orc.vector1D(0.0, 2.5, L) == -2.5
    assert orc.vector1D(2.5, 0.0, L) == 2.5
    # valid (one wrap) up to 1.5 L separation, e.g. an atom slightly outside the box
    assert orc.vector1D(-0.3, 4.9, L) == pytest.approx(0.2, abs=1e-15)
    rng = np.random.default_rng(0)
    for _ in range(2000):
        c1, c2 = rng.random(2) * L
        d = orc.vector1D(c1, c2, L)
        assert abs(d) <= L / 2 + 1e-15
        assert abs(((c2 - c1) - d) / L - round(((c2 - c1) - d) / L)) < 1e-12
        from metropolismontecarlo_amd.api import vector1D
        assert vector1D(c1, c2, L) == d


def test_branch_free_minimum_image_is_bit_identical():
    """The kernels evaluate vector1D as d - copysign(box, d) under |d| < box - |d|
    (csrc/mmc_device.hpp); it must equal the reference's two-branch form bit for bit, ties and
    signed zeros included."""
    rng = np.random.default_rng(1)
    L = 30.0
    c1 = np.concatenate([rng.random(20000) * L, [0.0, 0.0, 15.0, 7.5, 30.0, -0.0, 3.0, 1e-300]])
    c2 = np.concatenate([rng.random(20000) * L, [15.0, 30.0, 0.0, 22.5, 0.0, 0.0, 3.0, -1e-300]])
    c1 = np.concatenate([c1, c1 - 1.2, c1 + 0.7])          # atoms may sit slightly outside the box
    c2 = np.concatenate([c2, c2 + 0.9, c2 - 1.1])
    d = c2 - c1
    ad = np.abs(d)
    fast = np.where(ad < (L - ad), d, d - np.copysign(L, d))
    ref = np.array([orc.vector1D(float(x), float(y), L) for x, y in zip(c1, c2)])
    assert np.array_equal(fast, ref)
    assert np.array_equal(np.signbit(fast), np.signbit(ref))


def test_prepare_ewald_kvectors():
    ew = orc.Ewald(5.6 / 30.0, 5, 27, 30.0)
    assert ew.NKVECS == 337                       # strict k^2 < 27 (ewalds.jl:61,76), not 353
    k2 = (ew.kxyz.astype(int) ** 2).sum(1)
    assert k2.min() >= 1 and k2.max() <= 26
    assert (ew.kxyz[:, 0] >= 0).all()
    # the reference's nesting order: kx outer, ky, kz inner (ewalds.jl:71-73)
    key = (ew.kxyz[:, 0] * 11 + ew.kxyz[:, 1] + 5) * 11 + ew.kxyz[:, 2] + 5
    assert (np.diff(key) > 0).all()
    # cfac (ewalds.jl:79-83), doubled for kx > 0; the Gaussian factor is box independent (kL=5.6)
    kr2 = (2 * np.pi) ** 2 * k2
    c = 2 * np.pi * np.exp(-kr2 / 4 / 5.6 ** 2) / kr2 / 30.0 * np.where(ew.kxyz[:, 0] > 0, 2, 1)
    assert np.allclose(ew.cfac, c, rtol=1e-14)
    with pytest.raises(AssertionError):
        orc.Ewald(0.2, 5, 26, 30.0)


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_nist_fourier_and_self(k):
    """E_fourier, E_self: same formula as RecipLong*factor / EwaldSelf -> exact known answers."""
    a = common.nist_arrays(k)
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    e = orc.recip_long(ew, s.coords, s.charge, s.box) * ew.factor
    assert rel(e, common.NIST[k]["fourier"]) < 5e-6      # 6 printed digits
    assert rel(orc.ewald_self(ew, s.charge), common.NIST[k]["self"]) < 5e-6
    assert np.array_equal(ew.sumQExpOld, ew.sumQExpNew)  # ewalds.jl:600-601


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_nist_real_and_disp_pin_pair_arithmetic(k):
    """E_real = factor/2 * sum_i EwaldReal_atomcut(i) (Ewald/ewald.jl:124-169) and
    E_disp = potential() over the oxygens with the monatomic LJ_dU (Ewald/energy.jl:294-364)."""
    a = common.nist_arrays(k)
    s = common.oracle_system(a)
    kappa = 5.6 / s.box
    real = sum(orc.ewald_real_atomcut(i, s, kappa, RCUT) for i in range(1, s.n_mol + 1))
    assert rel(real * orc.factor() / 2, common.NIST[k]["real"]) < 5e-6
    from metropolismontecarlo_amd import io as mio
    ox = s.coords[0::3]
    e, _ = orc.potential_monatomic(ox, np.full(len(ox), mio.SPCE_EPS_O),
                                   np.full(len(ox), mio.SPCE_SIGMA_O), RCUT, s.box)
    assert rel(e, common.NIST[k]["disp"]) < 5e-6


def LennardJones(rij):  # Monatomic/mainMonatomic.jl:327-329
    return 4 * 1 * ((1 / rij) ** 12 - (1 / rij) ** 6)


def test_reference_test_LJ():
    """Monatomic/mainMonatomic.jl:292-325 (the only reference test that actually runs)."""
    box, r_cut = 5.0, 2.5
    r = np.array([[0, 0, 0], [0, 0, 2], [0, 1.5, 0]], dtype=float)
    enn, _ = orc.lj_du_monatomic(1, r, np.ones(3), np.ones(3), r_cut, box)
    assert abs(enn - (LennardJones(2.0) + LennardJones(1.5))) < 1e-3
    assert enn == pytest.approx(-0.38186003177857464, abs=1e-14)
    r[1] = [0, 0, 4]   # its image is at distance 1: tests minimum image
    enn, _ = orc.lj_du_monatomic(1, r, np.ones(3), np.ones(3), r_cut, box)
    assert abs(enn - (LennardJones(1.0) + LennardJones(1.5))) < 1e-3
    assert enn == pytest.approx(-0.32033659427857464, abs=1e-14)
    # r^2 == r_c^2 is included (`if rij_sq > rcut_sq` skip, :249)
    r[1] = [0, 0, 2.5]
    e_at, _ = orc.lj_du_monatomic(1, r, np.ones(3), np.ones(3), r_cut, 1000.0)
    assert e_at == pytest.approx(LennardJones(2.5) + LennardJones(1.5), abs=1e-14)


def two_triangles():
    """Ewald/tests.jl:8-82: two 3-site molecules, the second shifted +2 in z, box 1000."""
    alpha2 = 75.0 * np.pi / 180.0 / 2.0
    db = np.array([[-np.sin(alpha2), 0.0, -np.cos(alpha2) / 3.0],
                   [0.0, 0.0, 2 * np.cos(alpha2) / 3.0],
                   [np.sin(alpha2), 0.0, -np.cos(alpha2) / 3.0]])
    ra = np.vstack([db, db + [0, 0, 2]])
    rm = np.array([db.mean(0), (db + [0, 0, 2]).mean(0)])   # COM with unit masses (test_COM)
    return dict(com=rm, first_atom=[1, 4], last_atom=[3, 6], coords=ra, atype=np.ones(6, int),
                charge=np.zeros(6), eps=np.ones((1, 1)), sig=np.ones((1, 1)), box=1000.0)


def test_reference_test_two_LJ_triangles():
    a = two_triangles()
    s = common.oracle_system(a)
    expected = sum(LennardJones(np.linalg.norm(s.coords[i] - s.coords[j]))
                   for i in range(3) for j in range(3, 6))
    calc, _ = orc.lj_poly_du(1, s, 500.0)
    assert abs(expected - calc) < 1e-4
    assert calc == pytest.approx(expected, rel=1e-13)


def test_reference_test_COM():
    # Ewald/tests.jl:86-102
    c = np.array([[1, 2, 3], [2, 3, 4], [0, 1, 2]], float)
    assert np.allclose(c.mean(0), [1, 2, 3])


@pytest.mark.parametrize("k,variant", [(1, "reference"), (1, "unwrapped"), (3, "reference"),
                                       (4, "reference")])
def test_against_independent_numpy_statement(k, variant):
    """oracle/numpy_check.py restates the molecular-cutoff LJ_poly_dU / EwaldReal / RecipLong /
    RecipMove with vectorised numpy + scipy.special.erfc, written separately from the C code."""
    a = common.nist_arrays(k, variant)
    s = common.oracle_system(a)
    kappa = 5.6 / s.box
    ew = orc.Ewald(kappa, 5, 27, s.box)
    for i in (1, 2, s.n_mol // 2, s.n_mol):
        p, v = orc.lj_poly_du(i, s, RCUT)
        pn, vn = npc.lj_poly_du(i, a, RCUT)
        assert rel(p, pn) < 1e-12 and rel(v, vn, abs(pn)) < 1e-12
        e, ov = orc.ewald_real(i, s, kappa, RCUT)
        en, ovn = npc.ewald_real(i, a, kappa, RCUT, 0.5)
        assert ov == ovn and rel(e, en) < 1e-12
    e = orc.recip_long(ew, s.coords, s.charge, s.box)
    en, Sn = npc.recip_long(ew.kxyz, ew.cfac, a["coords"], a["charge"], a["box"])
    assert rel(e, en) < 1e-12
    assert np.abs(Sn - ew.sumQExpNew).max() < 1e-11 * np.abs(Sn).max()
    mv = common.golden(k, variant)["moves"][0]
    i = mv["mol"]
    r_old = s.coords[3 * (i - 1):3 * i].copy()
    de = orc.recip_move(s.box, ew, r_old, np.array(mv["atoms_new"]), s.charge[3 * (i - 1):3 * i])
    den = npc.recip_move_delta(ew.kxyz, ew.cfac, ew.sumQExpOld, r_old, np.array(mv["atoms_new"]),
                               a["charge"][3 * (i - 1):3 * i], a["box"]) * ew.factor
    assert rel(de, den, 1.0) < 1e-9


@pytest.mark.parametrize("k,variant", [(k, v) for k in (1, 2, 3, 4)
                                       for v in ("reference", "unwrapped")])
def test_golden_regression(k, variant):
    a = common.nist_arrays(k, variant)
    g = common.golden(k, variant)
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    assert ew.NKVECS == g["nkvecs"] and ew.factor == g["factor"]
    t = orc.potential_ewald(s, ew, RCUT, RCUT)
    for key, val in g["totals_ewald"].items():
        assert t[key] == pytest.approx(val, rel=1e-13), key
    for i, pm in g["per_mol"].items():
        assert orc.lj_poly_du(int(i), s, RCUT) == pytest.approx(tuple(pm["lj"]), rel=1e-13)
        e, ov = orc.ewald_real(int(i), s, ew.kappa, RCUT)
        assert e == pytest.approx(pm["real"][0], rel=1e-13) and int(ov) == pm["real"][1]
    for mv in g["moves"]:
        i = mv["mol"]
        d, ov = orc.trial_move(i, s, ew, RCUT, RCUT, np.array(mv["com_new"]),
                               np.array(mv["atoms_new"]))
        assert int(ov) == mv["overlap"]
        assert np.allclose(d, mv["d"], rtol=1e-10, atol=1e-7)
        if mv["accept"]:
            s.com[i - 1] = mv["com_new"]
            s.coords[3 * (i - 1):3 * i] = mv["atoms_new"]
            ew.sumQExpOld = ew.sumQExpNew.copy()
        else:
            ew.sumQExpNew = ew.sumQExpOld.copy()
        assert rel(np.abs(ew.sumQExpOld).sum(), mv["sum_abs_S_old"]) < 1e-12


def test_running_total_vs_recompute():
    """Poly/main.jl:232-235: total + accepted deltas == full recompute, and the incrementally
    updated S(k) == a fresh RecipLong."""
    a = common.nist_arrays(1, "unwrapped")
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    total = orc.potential_ewald(s, ew, RCUT, RCUT)["energy"]
    rng = np.random.default_rng(5)
    n_acc = 0
    for n in range(150):
        i = n % s.n_mol + 1
        d = (rng.random(3) - 0.5) * 0.316555789
        cn, an = s.com[i - 1] + d, s.coords[3 * (i - 1):3 * i] + d
        dd, ov = orc.trial_move(i, s, ew, RCUT, RCUT, cn, an)
        delta = dd[0] + dd[1] + dd[2]
        if not ov and (delta < 0 or np.exp(-delta / 298.15) > rng.random()):
            total += delta
            s.com[i - 1], s.coords[3 * (i - 1):3 * i] = cn, an
            ew.sumQExpOld = ew.sumQExpNew.copy()
            n_acc += 1
        else:
            ew.sumQExpNew = ew.sumQExpOld.copy()
    assert n_acc > 30
    S_inc = ew.sumQExpOld.copy()
    fresh = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    assert rel(total, orc.potential_ewald(s, fresh, RCUT, RCUT)["energy"]) < 1e-11
    assert np.abs(S_inc - fresh.sumQExpOld).max() < 1e-10 * np.abs(S_inc).max()


def test_overlap_and_quirks():
    a = common.nist_arrays(1, "unwrapped")
    a["coords"][3 * 6 + 1] = a["coords"][3 * 2] + np.array([0.3, 0.0, 0.0])  # H on top of an O
    s = common.oracle_system(a)
    kappa = 5.6 / s.box
    assert orc.ewald_real(3, s, kappa, RCUT) == (0.0, True)          # ewalds.jl:359-360
    assert orc.ewald_real(7, s, kappa, RCUT) == (0.0, True)
    # 0.3^2 = 0.09 < 0.5 < 1.0: both ovr conventions flag it; at 0.8 A only the legacy one does
    a["coords"][3 * 6 + 1] = a["coords"][3 * 2] + np.array([0.8, 0.0, 0.0])
    s = common.oracle_system(a)
    assert orc.ewald_real(3, s, kappa, RCUT, ovr=0.5)[1] is False
    assert orc.ewald_real(3, s, kappa, RCUT, ovr=1.0)[1] is True
    # same-sign close contact is NOT an overlap
    a["coords"][3 * 6] = a["coords"][3 * 2] + np.array([0.3, 0.0, 0.0])
    a["coords"][3 * 6 + 1] = a["coords"][3 * 6] + np.array([5.0, 0.0, 0.0])
    a["coords"][3 * 6 + 2] = a["coords"][3 * 6] + np.array([0.0, 5.0, 0.0])
    s = common.oracle_system(a)
    assert orc.ewald_real(3, s, kappa, RCUT)[1] is False
    with pytest.raises(AssertionError):
        orc.coulomb_real(1, s, 9.0)                                  # energy.jl:648
    ew = orc.Ewald(kappa, 5, 27, s.box)
    with pytest.raises(AssertionError):
        orc.recip_move(s.box, ew, np.zeros((2, 3)), np.zeros((2, 3)), np.zeros(2))


def test_wolf_prefactor_closed_form():
    """energy.jl:924-930 literally (O(N^2)) vs -(sum q)^2 erfc(kappa r_c)/r_c."""
    a = common.nist_arrays(1)
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    lit = orc.potential_wolf(s, ew, RCUT, RCUT, literal_prefactor=True)
    fast = orc.potential_wolf(s, ew, RCUT, RCUT, literal_prefactor=False)
    assert rel(lit["energy"], fast["energy"]) < 1e-13
    assert lit["virial"] == fast["virial"]


def test_kx0_plane_of_the_structure_factor_comes_in_exact_conjugate_pairs():
    """The reference keeps BOTH (0, ky, kz) and (0, -ky, -kz) (ewalds.jl:57-89 halves only kx > 0 by
    doubling cfac): 88 of the 337 k-vectors, 44 pairs.  Its phase tables take negative indices as
    conjugates (ewalds.jl:575-585, :781-796), and products and sums of conjugates are conjugates
    bit for bit in IEEE arithmetic: S(0,-ky,-kz) == conj(S(0,ky,kz)) EXACTLY, after RecipLong and
    after every RecipMove.  (the 13 % of S(k) traffic a layout without the
    redundant half would save.)"""
    from oracle import oracle as orc
    a = common.nist_arrays(2, "unwrapped")
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    orc.recip_long(ew, s.coords, s.charge, s.box)
    k = ew.kxyz.tolist()
    idx = {tuple(v): i for i, v in enumerate(k)}
    pairs = [(i, idx[(0, -v[1], -v[2])]) for i, v in enumerate(k) if v[0] == 0 and idx[(0, -v[1], -v[2])] > i]
    assert len(pairs) == 44 and sum(v[0] == 0 for v in k) == 88

    def exact(S):
        return all(S[j].real == S[i].real and S[j].imag == -S[i].imag for i, j in pairs)
    assert exact(ew.sumQExpNew) and exact(ew.sumQExpOld)
    assert all(ew.cfac[i] == ew.cfac[j] for i, j in pairs)
    rng = np.random.default_rng(3)
    for m in (1, 57, 200):
        d = (rng.random(3) - 0.5) * 0.3
        orc.trial_move(m, s, ew, 10.0, 10.0, s.com[m - 1] + d, s.coords[3 * m - 3:3 * m] + d)
        assert exact(ew.sumQExpNew)
        ew.sumQExpOld = ew.sumQExpNew.copy()
