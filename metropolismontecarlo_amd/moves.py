"""Host-side move generation and bookkeeping of the reference -- the callers of the hot path.

SURVEY.md section 8(f) rows 1 and 4: what `Loop()` does between the energy calls.  These are plain
host functions (numpy), written from the cited lines; `rng` is anything with `.random()`
(numpy Generator) -- the reference draws from Julia's global RNG, which it never seeds
(Ewald/main.jl:36), so no stream parity exists to preserve.

The native driver (csrc/mmc_engine.inc) carries its own C++ versions of `PBC`,
`random_translate_vector`, `random_vector` and `Metropolis`; it rotates the atom offsets rigidly
instead of going through `q_to_a`, so quirk Q12 below has no counterpart there.
"""
import math

import numpy as np


# ---- Ewald/boundaries.jl:16-26 -------------------------------------------------------------------
def PBC(v, box):
    x, y, z = (float(c) for c in v)
    if x > box: x -= box
    if x < 0: x += box
    if y > box: y -= box
    if y < 0: y += box
    if z > box: z -= box
    if z < 0: z += box
    return np.array([x, y, z])


# ---- Ewald/auxillary.jl:94-114 -------------------------------------------------------------------
def random_translate_vector(dr_max, old, box, rng):
    zeta = np.array([rng.random(), rng.random(), rng.random()]) - 0.5   # (-1/2, +1/2)
    return PBC(np.asarray(old, dtype=float) + zeta * dr_max, box)


def Metropolis(delta, rng):
    if delta < 0.0:
        return True
    return math.exp(-delta) > rng.random()


# ---- Ewald/auxillary.jl:145-178 ------------------------------------------------------------------
def COM(atoms, masses):
    atoms = np.asarray(atoms, dtype=float)
    masses = np.asarray(masses, dtype=float)
    return (atoms * masses[:, None]).sum(0) / masses.sum()


Center_of_Mass = COM


def MATMUL(ai, db):
    """auxillary.jl:154-159: (dot(db, ai[:,1]), dot(db, ai[:,2]), dot(db, ai[:,3])) = ai^T db."""
    ai = np.asarray(ai, dtype=float)
    db = np.asarray(db, dtype=float)
    return np.array([db @ ai[:, 0], db @ ai[:, 1], db @ ai[:, 2]])


# ---- Ewald/quaternions.jl ------------------------------------------------------------------------
def q_to_a(q, faithful=True):
    """quaternions.jl:11-50.  `faithful=True` keeps the reference's element (2,3):
    2*(q2*q4 + q1*q2) (a typo for 2*(q3*q4 + q1*q2), quirk Q12) -- the returned matrix is then
    not orthogonal.  faithful=False gives the Allen & Tildesley rotation matrix."""
    q = np.asarray(q, dtype=float)
    norm = q @ q
    if abs(norm - 1.0) > 1.0e-6:
        raise ValueError(f"quaternion normalization error {norm}")   # the reference exit()s
    q1, q2, q3, q4 = q
    a23 = 2 * (q2 * q4 + q1 * q2) if faithful else 2 * (q3 * q4 + q1 * q2)
    return np.array([
        [q1 ** 2 + q2 ** 2 - q3 ** 2 - q4 ** 2, 2 * (q2 * q3 + q1 * q4), 2 * (q2 * q4 - q1 * q3)],
        [2 * (q2 * q3 - q1 * q4), q1 ** 2 - q2 ** 2 + q3 ** 2 - q4 ** 2, a23],
        [2 * (q2 * q4 + q1 * q3), 2 * (q3 * q4 - q1 * q2), q1 ** 2 - q2 ** 2 - q3 ** 2 + q4 ** 2]])


def random_vector(rng):
    """quaternions.jl:52-74: uniform unit vector by rejection from the cube."""
    while True:
        e = 2.0 * np.array([rng.random(), rng.random(), rng.random()]) - 1.0
        norm = e @ e
        if norm < 1.0:
            break
    return e / math.sqrt(norm)


def quatmul(a, b):
    """quaternions.jl:76-91"""
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[1] * b[0] + a[0] * b[1] - a[3] * b[2] + a[2] * b[3],
                     a[2] * b[0] + a[3] * b[1] + a[0] * b[2] - a[1] * b[3],
                     a[3] * b[0] - a[2] * b[1] + a[1] * b[2] + a[0] * b[3]])


def rotate_quaternion(angle, axis, old):
    """quaternions.jl:93-120"""
    axis = np.asarray(axis, dtype=float)
    if abs(axis @ axis - 1.0) > 1.0e-6:
        raise ValueError("axis normalization error")
    rot = np.empty(4)
    rot[0] = math.cos(0.5 * angle)
    rot[1:] = math.sin(0.5 * angle) * axis
    return quatmul(rot, old)


def random_quaternion(rng):
    """quaternions.jl:122-156: uniform unit quaternion (Marsaglia)."""
    while True:
        z = 2.0 * np.array([rng.random(), rng.random()]) - 1.0
        norm1 = z @ z
        if norm1 < 1.0:
            break
    e0, e1 = z
    while True:
        z = 2.0 * np.array([rng.random(), rng.random()]) - 1.0
        norm2 = z @ z
        if norm2 < 1.0:
            break
    f = math.sqrt((1.0 - norm1) / norm2)
    return np.array([e0, e1, z[0] * f, z[1] * f])


def random_rotate_quaternion(angle_max, old, rng):
    """quaternions.jl:158-182"""
    old = np.asarray(old, dtype=float)
    if abs(old @ old - 1.0) > 1.0e-6:
        raise ValueError("old normalization error")
    axis = random_vector(rng)
    angle = (2.0 * rng.random() - 1.0) * angle_max
    return rotate_quaternion(angle, axis, old)


def space_fixed_atoms(com, q, db, faithful=True):
    """main.jl:543-549: ra[a] = COM + MATMUL(q_to_a(e), db[a]) for every body-fixed site."""
    ai = q_to_a(q, faithful)
    return np.array([np.asarray(com, dtype=float) + MATMUL(ai, site) for site in db])


# ---- Ewald/adjust.jl:1-41 (Adjust!) and :43-83 (Adjust_rot!, same body) -----------------------------
def Adjust(saved, L):
    """Frenkel-Smit step-size controller; `saved` is structs.Moves."""
    if saved.attempp == 0:
        saved.naccepp = saved.naccept
        saved.attempp = saved.attempt
    else:
        ratio = float(saved.naccept - saved.naccepp) / float(saved.attempt - saved.attempp)
        dr_old = saved.d_max
        saved.d_max = saved.d_max * ratio / saved.set_value
        dr_ratio = saved.d_max / dr_old
        if dr_ratio > 1.5:
            saved.d_max = dr_old * 1.5
        if dr_ratio < 0.5:
            saved.d_max = dr_old * 0.5
        if saved.d_max > L / 2:
            saved.d_max = L / 2
        saved.naccepp = saved.naccept
        saved.attempp = saved.attempt
    return saved


Adjust_rot = Adjust


# ---- Ewald/auxillary.jl:116-123 ------------------------------------------------------------------
def Pressure(vir, rho, T, vol):
    return rho * T + vir.virial / vol
