"""Host-side loaders that turn the reference's input decks into the flat arrays the hot path reads.

Only what the parity fixtures and the bench need: the NIST SPC/E sample-configuration reader
(Ewald/initialConfigurations.jl:282-355 `ReadNIST`) and the "nist" set-up branch of the driver
(Ewald/main.jl:231-275).  Loaders are outside the hot path; they run on
the host with numpy.
"""
import numpy as np

from .structs import Tables

# Ewald/main.jl:242-245
SPCE_SIGMA_O = 0.316555789 * 10.0  # nm -> Angstrom
SPCE_EPS_O = 78.1974311            # K
SPCE_Q_H = 0.42380                 # initialConfigurations.jl:329
SPCE_Q_O = -2 * 0.42380            # initialConfigurations.jl:316
NIST_MASS = (15.99, 1.009, 1.009)  # initialConfigurations.jl:344


def read_nist_text(path):
    """Parse a NIST SPC/E sample configuration: line 1 = box lengths, line 2 = N_mol, then
    `index x y z element` (initialConfigurations.jl:294-332).  Returns (box, xyz[n,3], is_oxygen[n])."""
    xyz, is_o = [], []
    box = None
    with open(path) as fh:
        for i, line in enumerate(fh, start=1):
            tok = line.split()
            if i == 1:
                box = float(tok[0])
            if len(tok) > 2 and i > 2:
                xyz.append([float(tok[1]), float(tok[2]), float(tok[3])])
                is_o.append(tok[4] == "O")
    return box, np.array(xyz, dtype=np.float64), np.array(is_o, dtype=bool)


def nist_system(box, xyz, is_oxygen, com="reference"):
    """Arrays of the SPC/E system as the reference's "nist" branch builds them.

    com="reference": centre of mass = mass-weighted mean of the RAW (individually wrapped) atom
        coordinates, exactly as ReadNIST does (initialConfigurations.jl:339-346) -- molecules that
        straddle the periodic boundary get a COM in the middle of the box.
    com="unwrapped": hydrogens are first brought next to their oxygen by minimum image, so every
        COM is physical; the COM is then wrapped into [0, L) carrying its atoms.
    Both then shift everything by |min COM| per axis (main.jl:247-275).
    Returns a dict with com, first_atom, last_atom (1-based inclusive), coords, atype (1-based),
    charge, eps, sig (Tables), box.
    """
    xyz = np.array(xyz, dtype=np.float64)
    n = xyz.shape[0]
    n_mol = n // 3
    assert n == 3 * n_mol and is_oxygen[0::3].all() and not is_oxygen[1::3].any()
    if com == "unwrapped":
        for m in range(n_mol):
            o = xyz[3 * m]
            for h in (1, 2):
                d = xyz[3 * m + h] - o
                xyz[3 * m + h] = o + d - box * np.round(d / box)
    elif com != "reference":
        raise ValueError(com)
    mass = np.array(NIST_MASS)
    rm = np.empty((n_mol, 3))
    for m in range(n_mol):  # COM(): sum(atoms .* masses) ./ totalMass  (auxillary.jl:145-150)
        a = xyz[3 * m:3 * m + 3]
        rm[m] = (a[0] * mass[0] + a[1] * mass[1] + a[2] * mass[2]) / (mass[0] + mass[1] + mass[2])
    shift = np.abs(rm.min(axis=0))  # main.jl:247-268
    rm = rm + shift
    ra = xyz + shift
    if com == "unwrapped":
        for m in range(n_mol):
            w = np.floor(rm[m] / box) * box
            rm[m] -= w
            ra[3 * m:3 * m + 3] -= w
    charge = np.where(is_oxygen, SPCE_Q_O, SPCE_Q_H)
    atype = np.where(is_oxygen, 1, 2).astype(np.int64)
    tab = Tables([SPCE_EPS_O, 0.0], [SPCE_SIGMA_O, 0.0])
    first = 3 * np.arange(n_mol, dtype=np.int64) + 1
    return dict(com=rm, first_atom=first, last_atom=first + 2, coords=ra, atype=atype,
                charge=charge, eps=tab.eps_ij, sig=tab.sig_ij, box=float(box))


def ReadNIST(filename):
    """initialConfigurations.jl:282-355: returns (qq_r, qq_q, rm, ra, atomTracker, box, atomName,
    atomType) -- raw, before the driver's shift."""
    box, xyz, is_o = read_nist_text(filename)
    n_mol = xyz.shape[0] // 3
    mass = np.array(NIST_MASS)
    rm = np.array([(xyz[3 * m] * mass[0] + xyz[3 * m + 1] * mass[1] + xyz[3 * m + 2] * mass[2])
                   / mass.sum() for m in range(n_mol)])
    qq_q = np.where(is_o, SPCE_Q_O, SPCE_Q_H)
    tracker = np.stack([3 * np.arange(n_mol) + 1, 3 * np.arange(n_mol) + 3], axis=1)
    names, num = [], 7
    for o in is_o:
        if o:
            num = 7
            names.append("O1")
        else:
            num += 1
            names.append("H" + str(num))
    atype = np.where(is_o, 1, 2)
    return xyz.copy(), qq_q, rm, xyz.copy(), tracker, box, names, atype


def cubic_lattice_water(n_mol, rho, geometry="spce", seed=11234):
    """Synthetic water box of the reference's crystal start (InitCubicGrid,
    initialConfigurations.jl:10-53: simple-cubic sites, offset 0.01 * spacing) with uniformly
    random orientations.  Used for the large synthetic configurations of BASELINE.json (cfg4/5).
    Returns (box, com[n_mol,3], coords[3 n_mol,3])."""
    box = (n_mol / rho) ** (1.0 / 3.0)
    nc = int(np.ceil(n_mol ** (1.0 / 3.0) - 1e-9))
    d = box / nc
    idx = np.arange(nc ** 3)[:n_mol]
    ix, iy, iz = idx // (nc * nc), (idx // nc) % nc, idx % nc
    com = (np.stack([ix, iy, iz], axis=1) + 0.01) * d
    if geometry == "spce":  # O-H 1.0 A, H-O-H 109.47 deg
        r_oh, ang = 1.0, np.deg2rad(109.47)
        masses = np.array([15.9994, 1.008, 1.008])
    elif geometry == "tip3p":
        # the three sites of Ewald/tip3p.pdb:3-5 as data (O, H, H; the file's geometry is
        # O-H 1.000 A, H-O-H 109.5 deg -- not the textbook TIP3P 0.9572 A / 104.52 deg)
        r_oh = ang = None
        masses = np.array([15.9994, 1.008, 1.008])
    else:
        raise ValueError(geometry)
    if r_oh is None:
        body = np.array([[-4.369, 0.061, -0.042], [-3.370, 0.049, 0.000], [-4.743, -0.180, 0.854]])
    else:
        body = np.array([[0.0, 0.0, 0.0],
                         [r_oh * np.sin(ang / 2), 0.0, r_oh * np.cos(ang / 2)],
                         [-r_oh * np.sin(ang / 2), 0.0, r_oh * np.cos(ang / 2)]])
    body = body - (body * masses[:, None]).sum(0) / masses.sum()
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n_mol, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.empty((n_mol, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    coords = com[:, None, :] + np.einsum("mij,aj->mai", R, body)
    return box, com, coords.reshape(-1, 3)
