"""The COM scan's fixed-point prefilter must never drop a neighbour the reference's gate accepts
(csrc/mmc_kernels.hpp: com_quant / com_quant_gate; Ewald/energy.jl:248-254, ewalds.jl:334-340).

The scan of k_move_eval_wave / k_total_wave tests 16-bit box fractions of the centres of mass and
gathers the full records of the survivors, to which the reference's own fp64 comparison is applied.
The construction below puts neighbours on both sides of the gate at relative distances down to
1e-12, across box faces and corners, with centres of mass outside the primary cell (the reference
wraps a difference once, not modulo L), and with the chosen molecule's proposal pushing
neighbours across the gate -- a dropped neighbour would change the result by 0.3 to 100 K, eight
orders of magnitude above the tolerance (1e-9 of the energy scale)."""
import numpy as np
import pytest

import common
from metropolismontecarlo_amd import io as mio
from metropolismontecarlo_amd import structs
from metropolismontecarlo_amd.device import Batch

pytestmark = pytest.mark.gpu

TOL = 1e-9
RCUT = 10.0


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def shell_system(box, centre, seed):
    """Molecule 1 at `centre`; the others on shells around it at r_cut (1 -+ eps), random
    directions and orientations, some shifted by a whole box."""
    rng = np.random.default_rng(seed)
    eps = [1e-12, 1e-10, 1e-8, 1e-6, 1e-4, 1e-2]
    com = [np.array(centre, dtype=float)]
    for e in eps:
        for sign in (-1.0, 1.0):
            for _ in range(9):
                u = rng.normal(size=3)
                u /= np.linalg.norm(u)
                c = com[0] + RCUT * (1.0 + sign * e) * u
                c -= box * np.floor(c / box)                    # into the primary cell ...
                if rng.random() < 0.3:
                    c += box * rng.integers(-1, 2, size=3)      # ... or one box beside it
                com.append(c)
    for _ in range(40):                                         # and ordinary neighbours / strangers
        com.append(rng.random(3) * box)
    com = np.array(com)
    # thin out overlapping molecules (an overlap is legal, but makes every later term moot)
    keep = [0]
    for k in range(1, len(com)):
        d = com[keep] - com[k]
        d -= box * np.rint(d / box)
        if (np.einsum("ij,ij->i", d, d) > 2.6 ** 2).all():
            keep.append(k)
    com = com[keep]
    n = len(com)
    _, _, body = mio.cubic_lattice_water(n, 0.03, "spce", seed=seed)   # random orientations
    body = body.reshape(n, 3, 3)
    lat_com = (body * np.array([15.9994, 1.008, 1.008])[None, :, None]).sum(1) / 18.0154
    coords = (body - lat_com[:, None, :] + com[:, None, :]).reshape(-1, 3)
    a4 = common.nist_arrays(4, "unwrapped")
    first = 3 * np.arange(n, dtype=np.int64) + 1
    return dict(com=com, coords=coords, first_atom=first, last_atom=first + 2,
                atype=np.tile([1, 2, 2], n).astype(np.int64),
                charge=np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], n), eps=a4["eps"],
                sig=a4["sig"], box=float(box))


@pytest.mark.parametrize("box,centre", [(30.0, (0.3, 29.8, 0.1)), (30.0, (15.0, 15.0, 15.0)),
                                         (30.0, (-0.05, 30.02, 14.0)), (24.7, (12.0, 0.01, 24.69)),
                                         (61.3, (61.29, 0.0, 30.0))])
def test_prefilter_is_a_superset_of_the_gate(box, centre, orc):
    a = shell_system(box, centre, seed=int(box * 10) + int(centre[0] * 7) % 5)
    n_mol = a["com"].shape[0]
    s_o = common.oracle_system(a)
    ew = orc.Ewald(5.6 / box, 5, 27, box)
    to = orc.potential_ewald(s_o, ew, RCUT, RCUT)
    orc.recip_long(ew, s_o.coords, s_o.charge, s_o.box)
    rng = np.random.default_rng(3)
    with Batch(2, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], box,
               5.6 / box, structs.factor, RCUT, RCUT) as b:
        b.set_option("kernel", 2)
        b.set_option("parts", 1)
        t = b.potential_ewald(as_array=True)                  # k_total_wave
        for key in ("energy", "virial", "lj", "real", "recip"):
            assert abs(t[key][0] - to[key]) < TOL * (abs(to[key]) + 1e4), (key, t[key][0], to[key])
        assert int(t["n_overlap"][0]) == int(to["n_overlap"])
        # trial moves of molecule 1 that carry the shells across the gate, and of shell molecules
        for step in range(12):
            i = 1 if step < 8 else int(rng.integers(2, n_mol + 1))
            d = (rng.random(3) - 0.5) * (2e-6 if step % 2 else 0.4)
            c_new = s_o.com[i - 1] + d
            a_new = s_o.coords[3 * (i - 1):3 * i] + d
            out, ov = b.eval(np.full(2, i), np.tile(c_new, (2, 1)), np.tile(a_new, (2, 1, 1)))
            do, ovo = orc.trial_move(i, s_o, ew, RCUT, RCUT, c_new, a_new)
            ew.sumQExpNew[:] = ew.sumQExpOld                  # every move is rejected
            assert bool(ov[0]) == ovo
            assert np.abs(out[0] - do).max() < TOL * (np.abs(do).max() + 1e4), (step, out[0], do)
        b.settle(np.zeros(2, dtype=bool))
