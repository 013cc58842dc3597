"""Host-side loaders that turn the reference's input decks into the flat arrays the hot path reads.

Only what the parity fixtures and the bench need: the NIST SPC/E sample-configuration reader
(Ewald/initialConfigurations.jl:282-355 `ReadNIST`) and the "nist" set-up branch of the driver
(Ewald/main.jl:231-275).  Loaders are outside the hot path; they run on
the host with numpy.
"""
import os

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


_NIST_NPZ = None


def load_nist_fixture(k, com="reference"):
    """NIST SPC/E sample configuration k = 1..4 (100, 200, 300 molecules in a 20 A box, 750 in
    30 A; configuration 4 is Ewald/coord750.txt) as nist_system() arrays.  The coordinates are
    public NIST data the reference redistributes; they ship with the package as
    data/spce_nist.npz (written by tests/golden/make_fixtures.py) because BASELINE's workload and
    the examples need them where /root/reference does not exist."""
    global _NIST_NPZ
    if _NIST_NPZ is None:
        _NIST_NPZ = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                                         "spce_nist.npz"))
    return nist_system(float(_NIST_NPZ[f"box_{k}"]), _NIST_NPZ[f"xyz_{k}"],
                       _NIST_NPZ[f"is_oxygen_{k}"], com)


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


def InitCubicGrid(n, rho):
    """Ewald/initialConfigurations.jl:10-53 (= Monatomic/mainMonatomic.jl:85-125): n sites of the
    smallest simple-cubic grid nCube^3 >= n (nCube >= 2) in a box of L = (n / rho)^(1/3), each at
    (index + 0.01) * L / nCube, x running fastest.  Returns (L, r[n, 3])."""
    L = (n / rho) ** (1.0 / 3.0)
    nCube = 2
    while nCube ** 3 < n:
        nCube += 1
    idx = np.arange(n)
    posit = np.stack([idx % nCube, (idx // nCube) % nCube, idx // (nCube * nCube)], axis=1)
    return L, (posit + 0.01) * (L / nCube)


def cubic_lattice_water(n_mol, rho, geometry="spce", seed=11234):
    """Synthetic water box of the reference's crystal start (InitCubicGrid,
    initialConfigurations.jl:10-53: simple-cubic sites, offset 0.01 * spacing) with uniformly
    random orientations.  Used for the large synthetic configurations of BASELINE.json (cfg4/5).
    Returns (box, com[n_mol,3], coords[3 n_mol,3])."""
    box, com = InitCubicGrid(n_mol, rho)
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


# ---- GROMACS-style input decks of the reference ---------------------------
R_GAS = 0.0083144621   # kJ/(mol K); Ewald/constants.jl -- vdwTable.eps /= R  (main.jl:185)


def ReadPDB(pdb_name):
    """Ewald/setup.jl:30-87.  The reference's fixed columns (1-based 31:38, 40:46, 48:55 for x, y,
    z; 12:15 atom name; 17:21 residue name; 22:27 residue number; 77:78 element), CRYST1 box.
    Returns a dict with the fields of the reference's `Topology` struct."""
    box = np.zeros(3)
    coords, atomnm, resnm, resnr, elem = [], [], [], [], []
    with open(pdb_name) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if "ATOM" in line or "HETATM" in line:
                coords.append([float(line[30:38]), float(line[39:46]), float(line[47:55])])
                atomnm.append(line[11:15].strip())
                resnm.append(line[16:21].strip())
                resnr.append(int(line[21:27]))
                elem.append(line[76:78].strip() if len(line) >= 77 else "")
            elif "CRYST1" in line:
                s = line.split()
                box = np.array([float(s[1]), float(s[2]), float(s[3])])
    name = os.path.basename(pdb_name).split(".")[0]
    return dict(name=name, box=box, r=np.array(coords, dtype=float).reshape(-1, 3), atomnm=atomnm,
                resnm=resnm, resnr=np.array(resnr, dtype=np.int64), elem=elem)


def ReadTopFile(top_file, substitutions=None):
    """Ewald/setup.jl:89-390: [ defaults ], [ atomtypes ], every [ moleculetype ] with its
    [ atoms ] (from the file itself or from `#include "x.itp"` files next to it) and
    [ molecules ].  Bonded sections are kept as raw token lists (nothing on the hot path reads
    them).  `substitutions` replaces template tokens such as the deck's `SOL SOLNUMBER`; without
    it a non-integer count raises like the reference's parse(Int64, ...).  Preprocessor lines
    (#ifndef / #else / #endif) are skipped like any other line without a section meaning, i.e.
    both branches are read -- the reference does the same."""
    substitutions = substitutions or {}
    top = dict(defaults=None, atomtypes=[], molparams=[], molecules={}, system=None)
    state = dict(zone="?", mol=None)

    def close_molecule():
        if state["mol"] is not None and state["mol"]["atoms"]:
            top["molparams"].append(state["mol"])
        state["mol"] = None

    def parse_lines(path):
        with open(path) as fh:
            for raw in fh:
                line = raw.split(";")[0].strip()
                if not line:
                    continue
                if line.startswith("#include"):
                    parse_lines(os.path.join(os.path.dirname(path), line.split('"')[1]))
                    continue
                if line.startswith("#"):
                    continue
                if line.startswith("[") and line.endswith("]"):
                    state["zone"] = line[1:-1].strip()
                    if state["zone"] in ("moleculetype", "system", "molecules"):
                        close_molecule()
                    continue
                s = line.split()
                z = state["zone"]
                if z == "defaults" and len(s) > 2:
                    top["defaults"] = dict(nbfunc=int(s[0]), comb_rule=int(s[1]), gen_pairs=s[2],
                                           fudgeLJ=float(s[3]), fudgeQQ=float(s[4]))
                elif z == "atomtypes" and len(s) > 2:
                    # name  bond_type  mass  charge  ptype  sigma(nm)  epsilon(kJ/mol)
                    top["atomtypes"].append(dict(name=s[0], atomicnr=s[1], mass=float(s[2]),
                                                 charge=float(s[3]), ptype=s[4], sigma=float(s[5]),
                                                 epsilon=float(s[6])))
                elif z == "moleculetype":
                    state["mol"] = dict(name=s[0], nrexcl=int(s[1]), atoms=[], bonds=[], pairs=[],
                                        angles=[], dihedrals=[])
                elif z == "atoms" and len(s) > 2:
                    state["mol"]["atoms"].append(dict(nr=int(s[0]), type=s[1], resnr=int(s[2]),
                                                      resnm=s[3], atomnm=s[4], cgnr=int(s[5]),
                                                      charge=float(s[6]), mass=float(s[7])))
                elif z in ("bonds", "pairs", "angles", "dihedrals") and state["mol"] is not None:
                    state["mol"][z].append(s)
                elif z == "system":
                    top["system"] = line
                elif z == "molecules":
                    tok = substitutions.get(s[1], s[1])
                    top["molecules"][s[0]] = int(tok)

    parse_lines(top_file)
    close_molecule()
    if len(top["molparams"]) != len(top["molecules"]):   # setup.jl:149-152
        raise ValueError(f"The number of moleculetypes in topology file {top_file} does not match "
                         "the number in section [ molecules ] of said file.")
    return top


def MakeTables(top):
    """Nonbonded part of Ewald/setup.jl:546-673 + the unit conversion of main.jl:185-186:
    Tables(eps, sig) mixing (structs.jl:337-347) over the [ atomtypes ] in file order, eps in K
    (kJ/mol / R), sig in Angstrom (nm * 10).  Returns structs.Tables."""
    from .structs import Tables
    eps = np.array([a["epsilon"] for a in top["atomtypes"]])
    sig = np.array([a["sigma"] for a in top["atomtypes"]])
    t = Tables(eps, sig)
    t.eps_ij = t.eps_ij / R_GAS
    t.sig_ij = t.sig_ij * 10.0
    return t


def system_from_decks(pdb, top):
    """MakeAtomArrays (setup.jl:393-445) reduced to what the hot path takes: for every PDB atom
    its molecule type (residue name == moleculetype name, or the PDB's residue name among the
    molecule's atom residue names), atom-type number (1-based index into [ atomtypes ]), charge
    and mass from the molecule's [ atoms ]; molecules from consecutive residue numbers; COM by
    mass.  Returns the arrays of Context.upload_system / Batch."""
    type_no = {a["name"]: k + 1 for k, a in enumerate(top["atomtypes"])}
    by_res = {}
    for m in top["molparams"]:
        by_res[m["name"]] = m
        for a in m["atoms"]:
            by_res.setdefault(a["resnm"], m)
    resnr = np.asarray(pdb["resnr"])
    resnr = resnr + (1 - resnr[0])                    # setup.jl:399-404
    starts = np.flatnonzero(np.r_[True, resnr[1:] != resnr[:-1]])
    first = starts + 1
    last = np.r_[starts[1:], len(resnr)]
    atype, charge, mass = [], [], []
    for f, l in zip(first, last):
        names = list(pdb["atomnm"][f - 1:l])
        m = by_res.get(pdb["resnm"][f - 1])
        if m is None:  # residue name unknown to the topology (mea.pdb: "MEA" vs "MEA_DUMMY"/"MOL"):
            cands = [x for x in top["molparams"] if [a["atomnm"] for a in x["atoms"]] == names]
            if len(cands) != 1:    # the reference resolves this through its global moleculeList
                raise KeyError(f"residue {pdb['resnm'][f - 1]!r} matches no moleculetype")
            m = cands[0]
        for nm in names:
            a = next(x for x in m["atoms"] if x["atomnm"] == nm)
            atype.append(type_no[a["type"]])
            charge.append(a["charge"])
            mass.append(a["mass"])
    mass = np.array(mass)
    r = pdb["r"]
    com = np.array([(r[f - 1:l] * mass[f - 1:l, None]).sum(0) / mass[f - 1:l].sum()
                    for f, l in zip(first, last)])
    tab = MakeTables(top)
    return dict(com=com, coords=r.copy(), first_atom=first.astype(np.int64),
                last_atom=last.astype(np.int64), atype=np.array(atype, dtype=np.int64),
                charge=np.array(charge), mass=mass, eps=tab.eps_ij, sig=tab.sig_ij, box=pdb["box"])


def PrintPDB(filename, step, coords, box, atom_names, mol_names, mol_numbers):
    """Ewald/initialConfigurations.jl:160-181: `<filename>_<step>.pdb` with the reference's
    CRYST1 / ATOM format strings (not strict PDB columns)."""
    box = np.broadcast_to(np.asarray(box, dtype=float), (3,))
    path = f"{filename}_{step}.pdb"
    with open(path, "w") as fh:
        fh.write("%-7s %7.3f %7.3f %7.3f %30s \n" % ("CRYST1", box[0], box[1], box[2],
                                                      "90.00  90.00  90.00 P 1           1"))
        for i, (xyz, an, mn, mol) in enumerate(zip(coords, atom_names, mol_names, mol_numbers), 1):
            fh.write("%-6s %4d %3s %4s %5d %3s %7.3f %7.3f %7.3f %5.2f %5.2f \n"
                     % ("ATOM", i, an, mn, mol, " ", xyz[0], xyz[1], xyz[2], 1.00, 0.00))
    return path
