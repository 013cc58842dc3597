"""Draw-level parity of the device-side move generator k_propose.

The generator is counter based: every uniform of (seed, replica, step) is Philox4x32-10 output the
library exports (mmc_philox4x32), so the exact move of any chain and step can be rebuilt on the
host.  Here it is rebuilt with the REFERENCE's formulas -- the host mirror moves.py of
random_translate_vector + PBC (Ewald/auxillary.jl:94-103, boundaries.jl:16-26), random_vector,
rotate_quaternion, random_rotate_quaternion, q_to_a (quaternions.jl:11-182) and
`ra = COM + MATMUL(ai, db[a])` (main.jl:543-549) -- fed with those uniforms in the order the
reference's functions consume them, and compared with the coordinates and quaternions the device
holds after one accepted step.

Tolerance: 2e-13 A on coordinates of magnitude <= 40 A (a few ulp: the device's sqrt / sincos and
the order of the three-term dot products differ from numpy's in the last bit), 4e-16 on quaternion
components; translations of the default (rigid) generator are bit-exact."""
import ctypes as C
import math

import numpy as np
import pytest

import common
from metropolismontecarlo_amd import _lib, moves
from metropolismontecarlo_amd import io as mio

pytestmark = pytest.mark.gpu

SLOT_KIND, SLOT_MOVE, SLOT_METROPOLIS, SLOT_AXIS = 0, 1, 2, 3


def philox_pair(seed, replica, step, slot):
    """The two uniforms of one slot, as mmc_draw (csrc/mmc_propose.hpp) forms them."""
    ctr = (C.c_uint32 * 4)(step & 0xFFFFFFFF, step >> 32, slot, replica)
    key = (C.c_uint32 * 2)(seed & 0xFFFFFFFF, seed >> 32)
    out = (C.c_uint32 * 4)()
    _lib.check(_lib.lib().mmc_philox4x32(ctr, key, out))
    a = (((out[0] << 32) | out[1]) >> 11) * 2.0 ** -53
    b = (((out[2] << 32) | out[3]) >> 11) * 2.0 ** -53
    return a, b


class ReferenceOrder:
    """`.random()` hands out the uniforms of one (seed, replica, step) in the order the
    reference's Loop() body consumes them: chose_move; then the three of random_translate_vector,
    or the triples of random_vector's rejection loop followed by the angle's zeta."""

    def __init__(self, seed, replica, step):
        self.k = lambda slot: philox_pair(seed, replica, step, slot)
        self.kind, self.zx = self.k(SLOT_KIND)
        self.mv = self.k(SLOT_MOVE)
        self.queue = None

    def chose_move(self):
        return self.kind

    def start_translation(self):
        self.queue = [self.zx, self.mv[0], self.mv[1]]

    def start_rotation(self):
        self.queue, slot = [], SLOT_AXIS
        while True:                       # the device draws two slots per axis attempt
            p, q = self.k(slot), self.k(slot + 1)
            slot += 2
            self.queue += [p[0], p[1], q[0]]
            e = 2.0 * np.array([p[0], p[1], q[0]]) - 1.0
            if e @ e < 1.0:
                break
        self.queue.append(self.mv[0])     # zeta of the angle

    def random(self):
        return self.queue.pop(0)


def water_body(tilted):
    """Body-fixed sites about the centre of mass.  tilted=False: SPC/E in its xz-plane (y = 0 for
    every site, so element (2,3) of q_to_a never meets a non-zero coordinate); tilted=True: the
    three sites of the reference's deck Ewald/tip3p.pdb:3-5 as they stand, all of x, y, z."""
    if tilted:
        body = np.array([[-4.369, 0.061, -0.042], [-3.370, 0.049, 0.000], [-4.743, -0.180, 0.854]])
    else:
        r_oh, ang = 1.0, math.radians(109.47)
        body = np.array([[0.0, 0.0, 0.0], [r_oh * math.sin(ang / 2), 0.0, r_oh * math.cos(ang / 2)],
                         [-r_oh * math.sin(ang / 2), 0.0, r_oh * math.cos(ang / 2)]])
    m = np.array([15.9994, 1.008, 1.008])
    return body - (body * m[:, None]).sum(0) / m.sum()


def quaternion_system(n_mol, faithful, seed=4, tilted=True):
    """Lattice COMs, random unit quaternions, atoms built the reference's way."""
    box, com = mio.InitCubicGrid(n_mol, 0.033101144)
    rng = np.random.default_rng(seed)
    quat = np.array([moves.random_quaternion(rng) for _ in range(n_mol)])
    db = water_body(tilted)
    coords = np.concatenate([moves.space_fixed_atoms(com[j], quat[j], db, faithful)
                             for j in range(n_mol)])
    a4 = common.nist_arrays(4, "unwrapped")
    return dict(com=com, coords=coords, atype=np.tile([1, 2, 2], n_mol).astype(np.int64),
                charge=np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], n_mol), eps=a4["eps"],
                sig=a4["sig"], box=float(box)), quat, db


def make_batch(a, R, rcut):
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Batch
    return Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
                 5.6 / a["box"], structs.factor, rcut, rcut)


@pytest.mark.parametrize("faithful", [True, False])
def test_quaternion_moves_match_the_reference_formulas_draw_by_draw(faithful):
    n_mol, R, seed, replica0 = 216, 96, 20260105, 1000
    dr_max, dphi_max = 0.4, 0.3
    a, quat, db = quaternion_system(n_mol, faithful)
    box = a["box"]
    with make_batch(a, R, 9.0) as b:
        b.set_option("device_moves", 1)
        b.set_orientations(quat, db, faithful=faithful)
        b.recip_long()
        n_t = n_r = n_rej = 0
        for step in range(3):             # three one-step calls: each moves molecule 1 (Loop()
                                          # restarts its sweep), the draw counter runs on
            before = [b.get_replica(r)[:2] for r in range(R)]
            q_before = [b.get_orientations(r) for r in range(R)]
            _, st = b.run(1, 1.0e12, dr_max, dphi_max, seed=seed, n_threads=2, replica0=replica0)
            i = 0                         # 0-based molecule of a call's first step (main.jl:490)
            for r in range(R):
                com, coords, _ = b.get_replica(r)
                q_now = b.get_orientations(r)
                com0, coords0 = before[r]
                others = np.arange(n_mol) != i
                assert np.array_equal(com[others], com0[others])
                assert np.array_equal(q_now[others], q_before[r][others])
                draws = ReferenceOrder(seed, replica0 + r, step)   # counter = steps run so far
                if draws.chose_move() < 0.5:                       # main.jl:519
                    draws.start_translation()
                    c_new = moves.random_translate_vector(dr_max, com0[i], box, draws)
                    e_new = q_before[r][i]
                    kind = 0
                else:
                    draws.start_rotation()
                    c_new = com0[i]
                    e_new = moves.random_rotate_quaternion(dphi_max, q_before[r][i], draws)
                    kind = 1
                at_new = moves.space_fixed_atoms(c_new, e_new, db, faithful)
                if np.array_equal(coords[3 * i:3 * i + 3], coords0[3 * i:3 * i + 3]):
                    n_rej += 1            # an overlap (the only way to be rejected at T -> inf)
                    assert np.array_equal(q_now[i], q_before[r][i])
                    continue
                n_t += kind == 0
                n_r += kind == 1
                assert np.array_equal(com[i], c_new), (r, step)       # bit-exact, both kinds
                assert np.abs(coords[3 * i:3 * i + 3] - at_new).max() < 2e-13, (r, step, kind)
                assert np.abs(q_now[i] - e_new).max() < 4e-16, (r, step, kind)
                assert (com[i] >= 0).all() and (com[i] <= box).all()
        assert n_t > 80 and n_r > 80 and n_rej < 0.05 * 3 * R
        # state consistency: every molecule is COM + MATMUL(q_to_a(quat), db), on the faithful or
        # the Allen-Tildesley matrix as selected
        com, coords, _ = b.get_replica(5)
        q_now = b.get_orientations(5)
        for j in range(n_mol):
            assert np.abs(coords[3 * j:3 * j + 3]
                          - moves.space_fixed_atoms(com[j], q_now[j], db, faithful)).max() < 2e-13
        bonds = np.linalg.norm(coords[0::3] - coords[1::3], axis=1)
        b0 = np.linalg.norm(db[0] - db[1])
        if faithful:   # quirk Q12: the reference's q_to_a is not a rotation -> molecules deform
            assert bonds.std() > 1e-3 * b0
        else:
            assert np.abs(bonds - b0).max() < 1e-12


def test_quaternion_mode_runs_a_consistent_chain():
    """Several sweeps in the reference's (faithful) mode: the running total energy equals a
    recompute, quaternions stay normalised within q_to_a's tolerance, and the run needs
    device-side proposals."""
    from metropolismontecarlo_amd._lib import MMCError
    n_mol, R = 125, 8
    a, quat, db = quaternion_system(n_mol, True, seed=8)
    with make_batch(a, R, 7.5) as b:
        b.set_orientations(quat, db, faithful=True)
        e0 = b.potential_ewald(as_array=True)["energy"].copy()
        with pytest.raises(MMCError, match="MMC_ERR_UNSUPPORTED"):
            b.run(1, 298.15, 0.3, 0.1, seed=1, energies=e0)       # host-side proposals
        b.set_option("device_moves", 1)
        e1, st = b.run(3 * n_mol, 298.15, 0.3, 0.1, seed=1, energies=e0, n_threads=2)
        t1 = b.potential_ewald(as_array=True)["energy"]
        assert np.abs(e1 - t1).max() < 1e-9 * np.abs(t1).max()
        assert st["rot_accept"] > 0 and st["trans_accept"] > 0
        q = b.get_orientations(R - 1)
        assert np.abs((q * q).sum(1) - 1.0).max() < 1e-6           # quaternions.jl:20
        assert np.abs(q - quat).max() > 1e-3
        bad = quat.copy()
        bad[3] *= 1.001
        with pytest.raises(AssertionError, match="quaternion normalization error"):
            b.set_orientations(bad, db)
        b.set_orientations(None, None)                             # back to the rigid generator
        e2, _ = b.run(50, 298.15, 0.3, 0.1, seed=2, energies=e1)
        assert np.abs(e2 - b.potential_ewald(as_array=True)["energy"]).max() < 1e-9 * np.abs(e2).max()


def test_rigid_generator_draw_by_draw():
    """The default generator (no quaternions): translations are random_translate_vector + PBC with
    the atoms following the wrapped centre of mass -- bit-exact against the host restatement;
    rotations turn the atoms about the centre of mass by random_vector's axis and the reference's
    uniform angle (quaternions.jl:52-74,176) -- compared at 2e-13."""
    a = common.nist_arrays(2, "unwrapped")
    n_mol, R, seed, box = a["com"].shape[0], 128, 77, a["box"]
    dr_max, dphi_max = 0.35, 0.25
    with make_batch(a, R, 10.0) as b:
        b.set_option("device_moves", 1)
        b.recip_long()
        b.run(1, 1.0e12, dr_max, dphi_max, seed=seed, n_threads=2, replica0=7)
        c0, x0 = a["com"][0], a["coords"][:3]
        n_t = n_r = 0
        for r in range(R):
            com, coords, _ = b.get_replica(r)
            if np.array_equal(coords[:3], x0):
                continue
            draws = ReferenceOrder(seed, 7 + r, 0)
            if draws.chose_move() < 0.5:
                draws.start_translation()
                c_new = moves.random_translate_vector(dr_max, c0, box, draws)
                assert np.array_equal(com[0], c_new)
                assert np.array_equal(coords[:3], x0 + (c_new - c0))
                n_t += 1
            else:
                draws.start_rotation()
                axis = moves.random_vector(draws)
                angle = (2.0 * draws.random() - 1.0) * dphi_max
                c, s = math.cos(angle), math.sin(angle)
                t = 1.0 - c
                ex, ey, ez = axis
                Rm = np.array([[t * ex * ex + c, t * ex * ey - s * ez, t * ex * ez + s * ey],
                               [t * ex * ey + s * ez, t * ey * ey + c, t * ey * ez - s * ex],
                               [t * ex * ez - s * ey, t * ey * ez + s * ex, t * ez * ez + c]])
                assert np.array_equal(com[0], c0)
                assert np.abs(coords[:3] - (c0 + (x0 - c0) @ Rm.T)).max() < 2e-13
                n_r += 1
        assert n_t > 40 and n_r > 40


@pytest.mark.parametrize("per_launch", [1, 8])
def test_quaternion_mode_with_the_kernel_s_own_decisions(per_launch):
    """The reference's quaternion route with the accept decision in the move kernel and several steps
    per launch: the same chains as with the host's decision -- energies, counts, coordinates AND
    the committed orientations (totProps.quat, main.jl:619)."""
    n_mol, R = 216, 8
    a, quat, db = quaternion_system(n_mol, True, seed=8)
    out = []
    for on_device in (0, 1):
        with make_batch(a, R, 7.5) as b:
            b.set_orientations(quat, db, faithful=True)
            b.set_option("device_moves", 1)
            b.set_option("kernel", 2)
            b.set_option("persistent", 0)
            b.set_option("accept_on_device", on_device)
            b.set_option("steps_per_launch", per_launch)
            e0 = b.potential_ewald(as_array=True)["energy"].copy()
            e1, st = b.run(2 * n_mol + 3, 298.15, 0.3, 0.1, seed=1, energies=e0, n_groups=2, n_parts=1,
                           n_threads=2)
            assert st["device_decisions"] == (st["moves"] if on_device else 0)
            t1 = b.potential_ewald(as_array=True)["energy"]
            assert np.abs(e1 - t1).max() < 1e-9 * np.abs(t1).max()
            out.append((e1.copy(), [st[k] for k in ("trans_accept", "rot_accept", "overlaps")],
                        b.get_orientations(R - 1), b.get_replica(R - 1)[:2]))
    assert out[0][1] == out[1][1] and out[0][1][0] > 0 and out[0][1][1] > 0
    assert np.array_equal(out[0][2], out[1][2])
    assert all(np.array_equal(x, y) for x, y in zip(out[0][3], out[1][3]))
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-12 * np.abs(out[0][0]).max()
