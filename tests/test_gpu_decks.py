"""The reference's TIP3P deck (tests/golden/decks: topol.top + tip3p.pdb) through the loaders onto
the device: per-molecule and total energies against the oracle with the deck's own parameters."""
import os

import numpy as np
import pytest

import common
from common import rel

pytestmark = pytest.mark.gpu

DECKS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decks")


def test_tip3p_deck_on_device_matches_oracle():
    from oracle import oracle as orc
    from metropolismontecarlo_amd import io as mio, structs
    from metropolismontecarlo_amd.device import Context
    top = mio.ReadTopFile(os.path.join(DECKS, "topol.top"), substitutions={"SOLNUMBER": 216})
    one = mio.system_from_decks(mio.ReadPDB(os.path.join(DECKS, "tip3p.pdb")), top)
    n_mol = top["molecules"]["SOL"]
    box, com, coords = mio.cubic_lattice_water(n_mol, 0.0331, "tip3p", seed=4)
    # only the water types interact here: restrict the 13-type table to (O1, H)
    a = dict(com=com, coords=coords, first_atom=3 * np.arange(n_mol, dtype=np.int64) + 1,
             last_atom=3 * np.arange(n_mol, dtype=np.int64) + 3,
             atype=np.tile(one["atype"], n_mol), charge=np.tile(one["charge"], n_mol),
             eps=one["eps"][:2, :2].copy(), sig=one["sig"][:2, :2].copy(), box=box)
    rc = 9.0
    assert rc < box / 2
    s = common.oracle_system(a)
    ew = orc.Ewald(5.6 / box, 5, 27, box)
    to = orc.potential_ewald(s, ew, rc, rc)
    with Context() as ctx:
        ctx.upload_system(a["com"], a["first_atom"], a["last_atom"], a["coords"], a["atype"],
                          a["charge"], a["eps"], a["sig"], box)
        ctx.prepare_ewald(5.6 / box, 5, 27, box, structs.factor)
        t = ctx.potential_ewald(rc, rc)
        for key in ("energy", "lj", "real", "recip", "self"):
            assert rel(t[key], to[key]) < 1e-9, key
        for i in (1, n_mol // 2, n_mol):
            e, v = ctx.lj_poly_du(i, rc)
            eo, vo = orc.lj_poly_du(i, s, rc)
            assert rel(e, eo, 1.0) < 1e-9 and rel(v, vo, 1.0) < 1e-9
