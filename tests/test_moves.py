"""CPU tests of the host-side move generation / bookkeeping mirror (moves.py), against the only
known answers the reference holds (Ewald/tests.jl:86-123) and the defining properties."""
import numpy as np
import pytest

from metropolismontecarlo_amd import moves
from metropolismontecarlo_amd.structs import Moves, Properties


def triatomic_db():
    # Ewald/tests.jl:168-176: isosceles triangle, unit bonds, 75 degrees; columns are sites
    a2 = 75.0 * np.pi / 180.0 / 2.0
    return np.array([[-np.sin(a2), 0.0, -np.cos(a2) / 3.0],
                     [0.0, 0.0, 2 * np.cos(a2) / 3.0],
                     [np.sin(a2), 0.0, -np.cos(a2) / 3.0]]).T


def test_reference_test_quaternion_matmul():
    # Ewald/tests.jl:104-123: "Calculated in Fortran as MATMUL(db(:,1),db)"
    db = triatomic_db()
    got = moves.MATMUL(db, db[:, 0])
    assert np.allclose(got, [0.440524936, -0.139868781, -0.300656140], atol=5e-9)


def test_reference_test_COM():
    # Ewald/tests.jl:86-102
    assert np.allclose(moves.COM([[1, 2, 3], [2, 3, 4], [0, 1, 2]], [1, 1, 1]), [1, 2, 3])
    assert np.allclose(moves.Center_of_Mass([[0, 0, 0], [1, 0, 0]], [3.0, 1.0]), [0.25, 0, 0])


def test_q_to_a_typo_and_corrected_form():
    rng = np.random.default_rng(3)
    q = moves.random_quaternion(rng)
    assert abs(q @ q - 1) < 1e-12
    good = moves.q_to_a(q, faithful=False)
    assert np.allclose(good @ good.T, np.eye(3), atol=1e-12) and np.linalg.det(good) == pytest.approx(1.0)
    ref = moves.q_to_a(q, faithful=True)
    diff = np.abs(ref - good)
    assert diff[1, 2] > 0 and np.count_nonzero(diff > 1e-15) == 1     # only element (2,3) differs
    assert ref[1, 2] == pytest.approx(2 * (q[1] * q[3] + q[0] * q[1]))  # quaternions.jl:43
    with pytest.raises(ValueError):
        moves.q_to_a([1.0, 1.0, 0.0, 0.0])
    assert np.allclose(moves.q_to_a([1, 0, 0, 0], faithful=True), np.eye(3))


def test_quaternion_algebra():
    rng = np.random.default_rng(4)
    a, b, c = (moves.random_quaternion(rng) for _ in range(3))
    assert np.allclose(moves.quatmul(moves.quatmul(a, b), c), moves.quatmul(a, moves.quatmul(b, c)))
    assert np.allclose(moves.quatmul([1, 0, 0, 0], a), a)
    axis = moves.random_vector(rng)
    assert abs(axis @ axis - 1) < 1e-12
    e = moves.rotate_quaternion(0.3, axis, a)
    assert abs(e @ e - 1) < 1e-12
    back = moves.rotate_quaternion(-0.3, axis, e)
    assert np.allclose(back, a)
    # the rotation taking a -> e (e * a^-1) has angle 0.3 about `axis`
    rel = moves.quatmul(e, a * np.array([1, -1, -1, -1]))
    assert rel[0] == pytest.approx(np.cos(0.15)) and np.allclose(rel[1:], np.sin(0.15) * axis)
    small = moves.random_rotate_quaternion(0.05, a, rng)
    ang = 2 * np.arccos(min(1.0, abs(moves.quatmul(small, a * np.array([1, -1, -1, -1]))[0])))
    assert ang <= 0.05 + 1e-12
    with pytest.raises(ValueError):
        moves.rotate_quaternion(0.1, [1, 1, 0], a)


def test_space_fixed_atoms_rigid_with_corrected_matrix():
    rng = np.random.default_rng(5)
    db = triatomic_db().T          # one row per site
    q = moves.random_quaternion(rng)
    com = np.array([3.0, 4.0, 5.0])
    ra = moves.space_fixed_atoms(com, q, db, faithful=False)
    d0 = np.linalg.norm(db[0] - db[1]), np.linalg.norm(db[0] - db[2])
    d1 = np.linalg.norm(ra[0] - ra[1]), np.linalg.norm(ra[0] - ra[2])
    assert np.allclose(d0, d1)
    assert np.allclose(moves.space_fixed_atoms(com, [1, 0, 0, 0], db), com + db)


def test_translation_and_pbc():
    rng = np.random.default_rng(6)
    box = 10.0
    for _ in range(200):
        old = rng.random(3) * box
        new = moves.random_translate_vector(0.3, old, box, rng)
        assert (new >= 0).all() and (new <= box).all()
        d = new - old
        d -= box * np.round(d / box)
        assert np.abs(d).max() <= 0.15 + 1e-12           # zeta in (-1/2, 1/2) times dr_max
    assert np.allclose(moves.PBC([10.2, -0.1, 5.0], 10.0), [0.2, 9.9, 5.0])
    assert np.allclose(moves.PBC([10.0, 0.0, 5.0], 10.0), [10.0, 0.0, 5.0])   # strict > and <


def test_metropolis():
    class R:
        def __init__(self, v): self.v = v
        def random(self): return self.v
    assert moves.Metropolis(-1.0, R(0.999))
    assert moves.Metropolis(1.0, R(0.3))          # exp(-1) = 0.3679 > 0.3
    assert not moves.Metropolis(1.0, R(0.4))
    assert not moves.Metropolis(0.0, R(1.0))      # exp(0) > 1.0 is false: delta == 0 draws


def test_adjust_controller():
    # Ewald/adjust.jl:1-41
    m = Moves(0, 40, 0, 100, 0.5, 0.2)
    moves.Adjust(m, 30.0)                          # first call only records
    assert (m.naccepp, m.attempp, m.d_max) == (40, 100, 0.2)
    m.naccept, m.attempt = 100, 200                # 60 % since last call -> d_max * 0.6/0.5
    moves.Adjust(m, 30.0)
    assert m.d_max == pytest.approx(0.24)
    m.naccept, m.attempt = 101, 300                # 1 % -> clamped to x0.5
    moves.Adjust(m, 30.0)
    assert m.d_max == pytest.approx(0.12)
    m.naccept, m.attempt = 201, 400                # 100 % -> clamped to x1.5
    moves.Adjust_rot(m, 30.0)
    assert m.d_max == pytest.approx(0.18)
    big = Moves(1, 101, 1, 101 + 0, 0.5, 14.0)
    big.attempt, big.naccept = 201, 201
    moves.Adjust(big, 30.0)
    assert big.d_max == 15.0                        # capped at L/2


def test_pressure():
    assert moves.Pressure(Properties(virial=30.0), 0.5, 2.0, 10.0) == pytest.approx(4.0)
