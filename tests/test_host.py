"""CPU tests of the host-side mirror: reference structs, loaders, bench bookkeeping."""
import numpy as np
import pytest

import common
from metropolismontecarlo_amd import io as mio
from metropolismontecarlo_amd import structs


def test_tables_mixing_rules():
    # Ewald/structs.jl:337-347 and the SPC/E values of main.jl:242-245
    t = structs.Tables([mio.SPCE_EPS_O, 0.0], [mio.SPCE_SIGMA_O, 0.0])
    assert t.eps_ij[0, 0] == pytest.approx(78.1974311) and t.eps_ij[0, 1] == 0 and t.eps_ij[1, 1] == 0
    assert t.sig_ij[0, 0] == pytest.approx(3.16555789) and t.sig_ij[0, 1] == pytest.approx(3.16555789 / 2)
    assert t.ϵᵢⱼ is t.eps_ij and t.σᵢⱼ is t.sig_ij        # the reference's field names
    t.ϵᵢⱼ = t.eps_ij / structs.R                           # main.jl:185 style rescale
    assert t.eps_ij[0, 0] == pytest.approx(78.1974311 / structs.R)


def test_ewald_struct_fields():
    ew = structs.EWALD(0.2, 5, 27, 2, [[1, 1, 1], [2, 2, 2]], [0.0, 0.0], np.zeros(2), np.zeros(2),
                       structs.factor)
    assert ew.kxyz.dtype == np.int32 and ew.sumQExpOld.dtype == np.complex128
    assert (ew.kappa, ew.nk, ew.k_sq_max, ew.NKVECS) == (0.2, 5, 27, 2)


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_nist_loader_conventions(k):
    ref = common.nist_arrays(k, "reference")
    unw = common.nist_arrays(k, "unwrapped")
    n = common.NIST[k]["n"]
    assert ref["com"].shape == (n, 3) and ref["coords"].shape == (3 * n, 3)
    assert ref["box"] == common.NIST[k]["L"]
    assert abs(ref["charge"].sum()) < 1e-9                 # main.jl:358 neutrality assert
    assert ref["charge"][0] == -0.8476 and ref["charge"][1] == 0.4238
    assert ref["com"].min() == pytest.approx(0.0, abs=1e-12)   # shifted by |min COM|, main.jl:247-271
    # quirk Q11: ReadNIST averages wrapped atoms -> some "molecules" are box-sized
    def oh(c):
        c = c.reshape(-1, 3, 3)
        return np.linalg.norm(c[:, 1] - c[:, 0], axis=1)
    broken = (oh(ref["coords"]) > 2.0).sum()
    assert broken > 0 and (oh(unw["coords"]) < 1.01).all()
    assert (oh(unw["coords"]) > 0.99).all()                # SPC/E O-H = 1.0 A
    assert (unw["com"] >= 0).all() and (unw["com"] < unw["box"]).all()
    # unbroken molecules are the same in both conventions up to the global shift
    ok = oh(ref["coords"]) < 2.0
    d = (unw["coords"].reshape(-1, 3, 3)[ok] - ref["coords"].reshape(-1, 3, 3)[ok])
    d -= np.round(d / ref["box"]) * ref["box"]
    assert np.abs(d - d[0, 0]).max() < 1e-9


def test_cubic_lattice():
    box, com, coords = mio.cubic_lattice_water(1000, 0.033101144)
    assert box == pytest.approx(31.1448, abs=1e-3)         # main.jl:117
    assert com.shape == (1000, 3) and coords.shape == (3000, 3)
    c = coords.reshape(-1, 3, 3)
    assert np.allclose(np.linalg.norm(c[:, 1] - c[:, 0], axis=1), 1.0)
    cosang = ((c[:, 1] - c[:, 0]) * (c[:, 2] - c[:, 0])).sum(1)
    assert np.allclose(np.degrees(np.arccos(cosang)), 109.47)


def test_bench_byte_model():
    import bench
    # 78.7 KB per move and 111.1 KB per full evaluation at 750 molecules, L = 30
    assert bench.algorithmic_bytes_per_move(750, 30.0) == pytest.approx(78.7e3, rel=2e-3)
    assert bench.algorithmic_bytes_full_eval(750) == pytest.approx(111.1e3, rel=2e-3)
    assert bench.algorithmic_bytes_per_move(1000, 31.1448) == pytest.approx(95.5e3, rel=5e-3)


def test_philox4x32_known_answers():
    """The driver's counter-based generator against the Random123 known-answer vectors
    (philox4x32, 10 rounds: zero, all-ones and the digits-of-pi counter/key)."""
    import ctypes as C
    from metropolismontecarlo_amd import _lib
    L = _lib.lib()

    def ph(ctr, key):
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        assert L.mmc_philox4x32(c, k, o) == 0
        return list(o)
    assert ph([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_branch_free_minimum_image_equals_reference_form_bitwise():
    """csrc/mmc_device.hpp vector1D: the reference's test `d < L - d` (ewalds.jl:30-38) is taken as
    |d| < L/2 and the wrapped value as d -+ L in one rounding.  Restated in numpy and compared
    bit for bit with the oracle's literal two-branch form, including ties at L/2, zero distance,
    images more than one box away and values one ulp either side of L/2."""
    from oracle import oracle as orc
    rng = np.random.default_rng(12)
    cases = []
    for box in (20.0, 30.0, 53.257, 67.09912345678, 1.0, 1000.0):
        c1 = rng.random(4000) * box
        c2 = rng.random(4000) * box
        cases += [(a, b, box) for a, b in zip(c1, c2)]
        h = box / 2
        for d in (h, np.nextafter(h, 0), np.nextafter(h, box), 0.0, box, 1.2 * box, 1.4999 * box):
            cases += [(0.0, d, box), (d, 0.0, box), (1.0, 1.0 + d, box), (1.0 + d, 1.0, box)]
    for c1, c2, box in cases:
        d = c2 - c1
        m = 0.0 if abs(d) < 0.5 * box else np.copysign(1.0, d)
        mine = d - box if m > 0 else d + box if m < 0 else d      # fma(m, -box, d), m in {0, +-1}
        ref = orc.vector1D(c1, c2, box)
        # the sign of a zero result is irrelevant (it is squared); everything else bit-identical
        assert mine == ref and (mine != 0.0 or ref == 0.0), (c1, c2, box, mine, ref)
        assert np.float64(mine).tobytes() == np.float64(ref).tobytes() or mine == 0.0


def test_rdf_restatement_and_normalisation_on_an_ideal_gas():
    """oracle/numpy_check.make_rdf_hist (gr.jl:60-91) + observables.normalize_rdf (gr.jl:92-104):
    every pair inside side/2 is counted once, and uniformly random points give g(r) ~ 1."""
    from oracle import numpy_check
    from metropolismontecarlo_amd import observables
    rng = np.random.default_rng(8)
    side, n, numbins = 10.0, 1500, 25
    pts = rng.random((n, 3)) * side
    hist = numpy_check.make_rdf_hist(pts, side, numbins)
    d = pts[:, None, :] - pts[None, :, :]
    d -= side * np.round(d / side)
    rr = np.sqrt((d ** 2).sum(-1))[np.triu_indices(n, 1)]
    assert hist.sum() == (rr <= side / 2).sum()
    assert hist[3] == ((rr > 2 * 0.2) & (rr <= 3 * 0.2)).sum()      # bin = ceil(r / dr), dr = 0.2
    r_, g = observables.normalize_rdf(hist, n, side, 1)
    assert np.allclose(r_[:2], [0.1, 0.3]) and abs(g[8:].mean() - 1.0) < 0.02


def test_block_line_is_the_references_format():
    """mmc_chain_block_line against the reference's @sprintf (Ewald/main.jl:667-679) applied to
    the same bookkeeping by hand: a host function, no GPU needed."""
    import numpy as np
    from metropolismontecarlo_amd._lib import CHAIN_DTYPE
    from metropolismontecarlo_amd.device import REFERENCE_IDEAL_TERM, block_line
    from metropolismontecarlo_amd import moves
    from metropolismontecarlo_amd.structs import Properties
    c = np.zeros(1, dtype=CHAIN_DTYPE)
    n_mol, box, blk = 750, 30.0, 7
    c["dr_max"], c["dphi_max"] = 0.3166, 0.0512
    c["energy"], c["virial"] = -17271541.515, -2.5e6
    c["avg_energy"], c["steps_taken"] = -17270000.0 * 15000, 15000
    c["overlaps"] = 3
    c["trans_naccept"], c["trans_attempt"] = 5800, 7512
    c["rot_naccept"], c["rot_attempt"] = 5621, 7488
    want = ("Block: %4d, Energy: %8.2f, Ratio trans: %4.2f, dr_max: %4.2f, Ratio rot: %4.2f, "
            "dϕ_max: %4.2f, instant energy: %8.2f, overlap count: %4d, pressure: %8.2f"
            % (blk, -17270000.0 / n_mol, 5800 / 7512, 0.3166, 5621 / 7488, 0.0512,
               -17271541.515 / n_mol, 3, 4.60453 + -2.5e6 / box / box / box))
    assert REFERENCE_IDEAL_TERM == 4.60453
    assert block_line(c[0], blk, n_mol, box) == want
    # with the ideal-gas term of auxillary.jl:121-123 instead of the literal
    rho, T = n_mol / box ** 3, 298.15
    line = block_line(c[0], blk, n_mol, box, ideal_term=rho * T)
    p = moves.Pressure(Properties(virial=-2.5e6), rho, T, box ** 3)
    assert line.endswith("pressure: %8.2f" % p)
    # no attempt yet: Julia prints NaN for 0/0
    c["rot_attempt"] = c["rot_naccept"] = 0
    assert "Ratio rot:  nan" in block_line(c[0], blk, n_mol, box).replace("NaN", "nan").replace("-nan", " nan")


def test_fixed_point_code_properties():
    """com_quant (csrc/mmc_kernels.hpp) restated in numpy: the wrapped 16-bit difference of two
    box-fraction codes exceeds the minimum-image distance by less than one unit per axis, which is
    what com_quant_gate's threshold relies on (the device side is tests/test_gpu_prefilter.py)."""
    rng = np.random.default_rng(0)
    L = 30.0
    x = rng.random((20000, 3)) * 3 * L - L
    y = x + (rng.random((20000, 3)) - 0.5) * L * 0.999       # |minimum image| < L / 2
    q = lambda v: (np.floor((v / L - np.floor(v / L)) * 65536.0).astype(np.int64)) & 0xFFFF
    D = ((q(x) - q(y) + 32768) % 65536) - 32768
    d = x - y
    d -= L * np.rint(d / L)
    u = L / 65536.0
    assert (np.abs(D) <= np.abs(d) / u + 1.0 + 1e-9).all()
