"""Loaders for the reference's GROMACS-style input decks against the decks
themselves -- tests/golden/decks/{tip3p.pdb, mea.pdb, topol.top} are the reference's data files
(Ewald/tip3p.pdb, mea.pdb, topol.top), kept here as input fixtures -- and the LJ tail corrections."""
import os

import numpy as np
import pytest

from metropolismontecarlo_amd import io as mio, npt

DECKS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decks")


def test_read_pdb_columns():
    t = mio.ReadPDB(os.path.join(DECKS, "tip3p.pdb"))
    assert t["name"] == "tip3p" and t["atomnm"] == ["O1", "H8", "H9"] and t["resnm"] == ["SOL"] * 3
    assert list(t["resnr"]) == [1, 1, 1] and t["elem"] == ["O", "H", "H"]
    assert np.allclose(t["r"], [[-4.369, 0.061, -0.042], [-3.370, 0.049, 0.000], [-4.743, -0.180, 0.854]])
    m = mio.ReadPDB(os.path.join(DECKS, "mea.pdb"))
    assert np.allclose(m["box"], [28.650, 28.930, 38.180]) and len(m["atomnm"]) == 11
    assert m["atomnm"][0] == "O2" and m["resnm"][0] == "MEA" and m["elem"][-1] == "H"
    assert np.allclose(m["r"][0], [21.7590, 20.9820, 22.7030])
    assert np.allclose(m["r"][-1], [21.7900, 21.5350, 18.8850])


def test_read_top_file_and_tables():
    path = os.path.join(DECKS, "topol.top")
    with pytest.raises(ValueError):                      # the deck's `SOL SOLNUMBER` placeholder
        mio.ReadTopFile(path)
    top = mio.ReadTopFile(path, substitutions={"SOLNUMBER": 1000})
    assert top["defaults"] == dict(nbfunc=1, comb_rule=2, gen_pairs="yes", fudgeLJ=0.5, fudgeQQ=0.8333)
    names = [a["name"] for a in top["atomtypes"]]
    assert names[:3] == ["O1", "H", "oh"] and len(names) == 13
    assert top["molecules"] == {"MEA_DUMMY": 1, "SOL": 1000}
    mea, sol = top["molparams"]
    assert mea["name"] == "MEA_DUMMY" and mea["nrexcl"] == 3 and len(mea["atoms"]) == 11
    assert len(mea["bonds"]) == 10 and len(mea["angles"]) == 16 and len(mea["dihedrals"]) == 18
    assert sum(a["charge"] for a in mea["atoms"]) == pytest.approx(1e-6, abs=2e-6)   # neutral
    assert [a["charge"] for a in sol["atoms"]] == [-0.834, 0.417, 0.417]
    assert [a["atomnm"] for a in sol["atoms"]] == ["O1", "H8", "H9"]
    tab = mio.MakeTables(top)
    # sigma_OO = 3.15061 A, eps_OO = 0.6364 / 0.0083144621 = 76.5413 K
    assert tab.sig_ij[0, 0] == pytest.approx(3.15061) and tab.eps_ij[0, 0] == pytest.approx(76.5413, rel=1e-6)
    assert tab.eps_ij[0, 1] == 0.0 and tab.sig_ij[0, 1] == pytest.approx(3.15061 / 2)   # O-H: eps 0
    oh, c1 = names.index("oh"), names.index("c1")
    assert tab.eps_ij[oh, c1] == pytest.approx(np.sqrt(0.880314 * 0.45773) / 0.0083144621)
    assert tab.sig_ij[oh, c1] == pytest.approx((3.06647 + 3.39967) / 2)


def test_system_from_decks_and_pdb_writer(tmp_path):
    top = mio.ReadTopFile(os.path.join(DECKS, "topol.top"), substitutions={"SOLNUMBER": 1})
    s = mio.system_from_decks(mio.ReadPDB(os.path.join(DECKS, "tip3p.pdb")), top)
    assert list(s["first_atom"]) == [1] and list(s["last_atom"]) == [3]
    assert list(s["atype"]) == [1, 2, 2] and list(s["charge"]) == [-0.834, 0.417, 0.417]
    m = np.array([15.9994, 1.008, 1.008])
    assert np.allclose(s["com"][0], (s["coords"] * m[:, None]).sum(0) / m.sum())
    s2 = mio.system_from_decks(mio.ReadPDB(os.path.join(DECKS, "mea.pdb")), top)
    assert list(s2["last_atom"]) == [11] and s2["charge"].sum() == pytest.approx(0.0, abs=2e-6)
    assert s2["atype"][0] == 3 and s2["eps"].shape == (13, 13)
    # writer: the reference's format strings, readable back with its own column reader
    path = mio.PrintPDB(str(tmp_path / "final"), 7, s["coords"], 30.0, ["O1", "H", "H"], ["SOL"] * 3,
                        [1, 1, 1])
    lines = open(path).read().splitlines()
    assert path.endswith("final_7.pdb") and lines[0].startswith("CRYST1   30.000  30.000  30.000")
    assert lines[1] == "ATOM      1  O1  SOL     1      -4.369   0.061  -0.042  1.00  0.00 "
    back = mio.ReadPDB(path)
    assert np.allclose(back["r"], s["coords"], atol=5e-4) and np.allclose(back["box"], 30.0)


def test_tail_corrections_match_the_textbook_single_component_form():
    # one atom type: U_tail = (8/3) pi N rho eps sig^3 [(1/3)(sig/rc)^9 - (sig/rc)^3],
    # P_tail = (16/3) pi rho^2 eps sig^3 [(2/3)(sig/rc)^9 - (sig/rc)^3]  (Allen & Tildesley 2.138-9
    # with the reference's prefactor convention: eps, not 4 eps -- energy.jl:556,606)
    n, eps, sig, rc, box = 750.0, 78.2, 3.166, 10.0, 30.0
    rho = n / box ** 3
    x3 = (sig / rc) ** 3
    assert npt.ener_corr([[eps]], [[sig]], rc, box, [n]) == pytest.approx(
        8.0 / 3.0 * np.pi * n * rho * eps * sig ** 3 * (x3 ** 3 / 3 - x3))
    assert npt.press_corr([[eps]], [[sig]], rc, box, [n]) == pytest.approx(
        16.0 / 3.0 * np.pi * rho ** 2 * eps * sig ** 3 * (2 * x3 ** 3 / 3 - x3))
    # two types, only O-O interacting: the H count must not matter
    e2 = npt.ener_corr([[eps, 0], [0, 0]], [[sig, sig / 2], [sig / 2, 0]], rc, box, [n, 2 * n])
    assert e2 == pytest.approx(npt.ener_corr([[eps]], [[sig]], rc, box, [n]))
