"""The reference's call surface for the hot path, served by libmmc_hip.so.

Same names, positional arguments, return values and error behaviour as the Julia methods that
`Loop()` (Ewald/main.jl:460-696) and `potential()` (Ewald/energy.jl:946-1032) call, so a parity
test reads like the reference's own code.  (`LJ_poly_ΔU` is a valid Python identifier.)  The Julia
module with the same methods over `ccall` is metropolismontecarlo_amd/julia/MMCHip.jl.

Device mirroring.  The reference keeps everything in host arrays that `Loop` mutates between calls
(moa.COM[i], soa.coords[first:last], ewald.sumQExpOld/New).  A *session* (one mmc_ctx) is attached
to each `soa`; before a per-molecule call the session re-sends molecule i and the molecule it sent
last time (which `Loop` may have restored on rejection, main.jl:623-624): that is exactly the set of
entries `Loop` can have changed.  Total-energy calls re-send all coordinates.  The structure-factor
arrays live on the device; `RecipMove`/`RecipLong` write the result back into `ewald.sumQExpNew`/
`sumQExpOld` and, before computing, push the host arrays if the caller rebound them (main.jl:621,628
rebind them to copies) -- detected by comparing their content with what the device was last
given (object identity is recycled by the allocator).  `sync_system(moa, soa)` forces a full
re-send after arbitrary host edits.

Nothing here computes energies on the host: without the HIP library and a GPU every function
raises.
"""
import weakref

import numpy as np

from . import structs
from .device import Context
from .structs import EWALD, Properties, Requirements, StructArray, Tables

__all__ = ["vector1D", "PrepareEwaldVariables", "LJ_poly_ΔU", "EwaldReal", "EwaldShort",
           "CoulombReal", "RecipLong", "RecipMove", "RecipCommit", "RecipRollback", "EwaldSelf",
           "potential", "sync_system", "release_sessions"]


def vector1D(c1, c2, box_size):
    """Ewald/ewalds.jl:30-38 == boundaries.jl:8-14 (host scalar helper; the kernels carry their
    own copy in mmc_device.hpp)."""
    if c1 < c2:
        return (c2 - c1) if (c2 - c1) < (c1 - c2 + box_size) else (c2 - c1 - box_size)
    return (c2 - c1) if (c1 - c2) < (c2 - c1 + box_size) else (c2 - c1 + box_size)


class _Session:
    def __init__(self, moa, soa, table, box, device=0):
        self.ctx = Context(device)
        nt = int(np.max(soa.atype))
        if table is None:
            eps = np.zeros((nt, nt))
            sig = np.zeros((nt, nt))
            self.has_table = False
        else:
            eps, sig = table.eps_ij, table.sig_ij
            self.has_table = True
        self.table_sig = None if table is None else (eps.tobytes(), sig.tobytes())
        self.ctx.upload_system(moa.COM, moa.firstAtom, moa.lastAtom, soa.coords, soa.atype,
                               soa.charge, eps, sig, box)
        self.box = float(box)
        self.last_mol = None
        self.ewald_key = None
        self.s_ids = (None, None)  # host copies of what the device S buffers hold
        self.moa = weakref.ref(moa)
        self._soa_coords_id = id(soa.coords)

    def sync_molecule(self, moa, soa, i):
        for m in {i, self.last_mol} - {None}:
            f, l = int(moa.firstAtom[m - 1]), int(moa.lastAtom[m - 1])
            self.ctx.set_molecule(m, moa.COM[m - 1], soa.coords[f - 1:l])
        self.last_mol = i

    def sync_all(self, moa, soa):
        self.ctx.update_system(moa.COM, soa.coords)
        self.last_mol = None

    def bind_ewald(self, ewald, box):
        key = (ewald.kappa, ewald.nk, ewald.k_sq_max, ewald.factor, float(box))
        if self.ewald_key != key:
            n = self.ctx.prepare_ewald(ewald.kappa, ewald.nk, ewald.k_sq_max, box, ewald.factor)
            self.ewald_key = key
            self.s_ids = (None, None)
            if ewald.NKVECS != n:  # a dummy EWALD as at main.jl:290-301
                ewald.NKVECS = n
        ewald._session = self

    def push_s(self, ewald):
        """Send sumQExpOld/New when the host arrays differ from what the device holds (Loop
        rebinds them to copies, main.jl:621,628).  Compared by CONTENT -- 337 complex numbers --
        because object identity can be recycled by the allocator."""
        so = np.asarray(ewald.sumQExpOld, dtype=np.complex128)
        sn = np.asarray(ewald.sumQExpNew, dtype=np.complex128)
        if len(so) != self.ctx.nkvecs or len(sn) != self.ctx.nkvecs:
            return
        m = self.s_ids
        if m[0] is None or not (np.array_equal(so, m[0]) and np.array_equal(sn, m[1])):
            self.ctx.set_sumqexp(so, sn)
            self.s_ids = (so.copy(), sn.copy())

    def pull_s(self, ewald, old=False):
        so, sn = self.ctx.get_sumqexp()
        if old:
            ewald.sumQExpOld = so
        ewald.sumQExpNew = sn
        self.s_ids = (so.copy(), sn.copy())


_sessions = {}


def _drop(key):
    s = _sessions.pop(key, None)
    if s is not None:
        s.ctx.close()


def _session(moa, soa, table, box):
    key = id(soa)
    s = _sessions.get(key)
    stale = s is not None and (s.box != float(box) or s.ctx.n_atoms != len(soa.coords)
                               or s.moa() is not moa
                               or (table is not None and not s.has_table)
                               or (table is not None and s.table_sig !=
                                   (table.eps_ij.tobytes(), table.sig_ij.tobytes())))
    if stale:
        _drop(key)
        s = None
    if s is None:
        s = _Session(moa, soa, table, box)
        _sessions[key] = s
        weakref.finalize(soa, _drop, key)
    return s


def release_sessions():
    """Destroy every device context this module created."""
    for k in list(_sessions):
        _drop(k)


def sync_system(moa, soa, box=None):
    """Re-send every COM and atom position of (moa, soa) -- after host edits that are not the
    single-molecule pattern of Loop()."""
    s = _sessions.get(id(soa))
    if s is not None:
        s.sync_all(moa, soa)


def _as_moa_soa(system):
    """Requirements (auxillary.jl:59-75) -> the moa/soa columns the kernels read."""
    cached = getattr(system, "_mmc_cache", None)
    tma = np.asarray(system.thisMol_theseAtoms, dtype=np.int64).reshape(-1, 2)
    if cached is None:
        moa = structs.make_moa(system.rm, tma[:, 0], tma[:, 1])
        soa = structs.make_soa(system.ra, system.atomTypes, np.zeros(len(system.ra)))
        system._mmc_cache = cached = (moa, soa)
    moa, soa = cached
    moa.COM[...] = np.asarray(system.rm, dtype=np.float64).reshape(-1, 3)
    soa.coords[...] = np.asarray(system.ra, dtype=np.float64).reshape(-1, 3)
    return moa, soa


# ------------------------------------------------------------------------------------------------
def PrepareEwaldVariables(ewald, boxSize):
    """Ewald/ewalds.jl:45-103 -> a NEW EWALD with kxyz, cfac, zeroed sumQExp arrays."""
    box = float(np.min(boxSize))
    with Context() as ctx:
        n = ctx.prepare_ewald(ewald.kappa, ewald.nk, ewald.k_sq_max, box, ewald.factor)
        kxyz, cfac = ctx.get_kvectors()
    return EWALD(ewald.kappa, ewald.nk, ewald.k_sq_max, n, kxyz, cfac,
                 np.zeros(n, dtype=np.complex128), np.zeros(n, dtype=np.complex128), ewald.factor)


def LJ_poly_ΔU(i, *args):
    """LJ_poly_ΔU(i, moa, soa, vdwTable, r_cut, box)   Ewald/energy.jl:209-290
    LJ_poly_ΔU(i, system::Requirements)              Ewald/energy.jl:126-206
    -> (energy, virial)."""
    if len(args) == 1:
        system = args[0]
        moa, soa = _as_moa_soa(system)
        vdwTable, r_cut, box = system.table, system.r_cut, system.box
        s = _session(moa, soa, vdwTable, box)
        s.sync_all(moa, soa)
    else:
        moa, soa, vdwTable, r_cut, box = args
        s = _session(moa, soa, vdwTable, box)
        s.sync_molecule(moa, soa, i)
    return s.ctx.lj_poly_du(i, r_cut)


def EwaldReal(*args):
    """EwaldReal(chosenOne, moa, soa, ewald, r_cut, box)                    Ewald/ewalds.jl:293-376
    EwaldReal(qq_r, qq_q, kappa, box, thisMol_thisAtom, chosenOne, system)  Ewald/ewalds.jl:205-289
    -> (pot, overlap), no factor."""
    if len(args) == 6:
        chosenOne, moa, soa, ewald, r_cut, box = args
        s = _session(moa, soa, None, box)
        s.bind_ewald(ewald, box)
        s.sync_molecule(moa, soa, chosenOne)
        return s.ctx.ewald_real(chosenOne, r_cut, 0.5)
    qq_r, qq_q, kappa, box, thisMol_thisAtom, chosenOne, system = args
    moa, soa = _as_moa_soa(system)
    soa.coords[...] = np.asarray(qq_r, dtype=np.float64).reshape(-1, 3)
    if not np.array_equal(soa.charge, qq_q):
        soa.charge = np.ascontiguousarray(qq_q, dtype=np.float64)
        _drop(id(soa))
    s = _session(moa, soa, system.table, box)
    s.bind_ewald(EWALD(kappa, 5, 27, 0, np.zeros((0, 3)), [], [], [], structs.factor), box)
    s.sync_all(moa, soa)
    return s.ctx.ewald_real(chosenOne, system.r_cut, 1.0)  # ovr = 1.0 (:240)


def EwaldShort(i, moa, soa, sim_props, ewald, box):
    """Ewald/ewalds.jl:892-910 -> (e, e/3, overlap), factor applied."""
    s = _session(moa, soa, None, box)
    s.bind_ewald(ewald, box)
    s.sync_molecule(moa, soa, i)
    return s.ctx.ewald_short(i, sim_props.qq_rcut)


def CoulombReal(qq_r, qq_q, box, chosenOne, system):
    """Ewald/energy.jl:618-711 (bare Coulomb) -> (pot, overlap)."""
    moa, soa = _as_moa_soa(system)
    soa.coords[...] = np.asarray(qq_r, dtype=np.float64).reshape(-1, 3)
    if not np.array_equal(soa.charge, qq_q):
        soa.charge = np.ascontiguousarray(qq_q, dtype=np.float64)
        _drop(id(soa))
    s = _session(moa, soa, system.table, box)
    s.sync_all(moa, soa)
    return s.ctx.coulomb_real(chosenOne, system.r_cut)


def _session_for_atoms(r, qq_q, box, ewald):
    """Session whose soa.coords is `r` (the usual call, energy.jl:1008), else an ad-hoc one in
    which every atom is its own molecule."""
    for s in _sessions.values():
        moa = s.moa()
        if moa is not None and s.ctx.n_atoms == len(r) and getattr(s, "_soa_coords_id", None) == id(r):
            return s
    holder = getattr(ewald, "_atoms_holder", None)
    if holder is None or holder[0] is not r:
        n = len(r)
        idx = np.arange(1, n + 1, dtype=np.int64)
        moa = structs.make_moa(np.array(r, dtype=np.float64).reshape(-1, 3), idx, idx)
        soa = structs.make_soa(np.asarray(r, dtype=np.float64).reshape(-1, 3), np.ones(n, np.int64),
                               qq_q)
        ewald._atoms_holder = holder = (r, moa, soa)
    _, moa, soa = holder
    soa.coords[...] = np.asarray(r, dtype=np.float64).reshape(-1, 3)
    moa.COM[...] = soa.coords
    s = _session(moa, soa, None, box)
    s.sync_all(moa, soa)
    return s


def RecipLong(*args):
    """RecipLong(ewald, r, qq_q, box)          Ewald/ewalds.jl:538-604
    RecipLong(system, ewald, r, qq_q)          Ewald/ewalds.jl:465-534
    -> (energy without factor, ewald); fills ewald.sumQExpOld and sumQExpNew."""
    if isinstance(args[0], Requirements):
        system, ewald, r, qq_q = args
        box = system.box
    else:
        ewald, r, qq_q, box = args
    s = getattr(ewald, "_session", None)
    if s is None or s.ctx.n_atoms != len(r) or s.box != float(box) or s.ctx._h is None:
        s = _session_for_atoms(r, qq_q, box, ewald)
    else:
        moa = s.moa()
        # all atoms matter here: re-send the coordinates the caller passed
        s.ctx.update_system(moa.COM, np.asarray(r, dtype=np.float64).reshape(-1, 3))
        s.last_mol = None
    s.bind_ewald(ewald, box)
    energy = s.ctx.recip_long()
    s.pull_s(ewald, old=True)
    return energy, ewald


def RecipMove(box, ewalds, r_old, r_new, qq_q):
    """Ewald/ewalds.jl:718-826 -> (energy * factor, ewalds); ewalds.sumQExpNew updated."""
    if len(r_old) != 3:
        raise AssertionError("n == 3 (ewalds.jl:740)")
    s = getattr(ewalds, "_session", None)
    if s is None or s.ctx._h is None:
        holder = structs.make_soa(np.zeros((1, 3)), [1], [0.0])
        s = _session(structs.make_moa(np.zeros((1, 3)), [1], [1]), holder, None, box)
        ewalds._standalone = holder
    if (ewalds.kappa, ewalds.nk, ewalds.k_sq_max, ewalds.factor, float(box)) != s.ewald_key:
        if ewalds.k_sq_max != 27:
            raise AssertionError("k_sq_max == 27 (ewalds.jl:742)")
        s.bind_ewald(ewalds, box)
    s.push_s(ewalds)
    de = s.ctx.recip_move(r_old, r_new, qq_q)
    s.pull_s(ewalds)
    return de, ewalds


def RecipCommit(ewald):
    """`ewald.sumQExpOld = [item for item in ewald.sumQExpNew]` (Ewald/main.jl:621) done on the
    device; keeps the host arrays in step."""
    s = ewald._session
    s.ctx.recip_commit()
    ewald.sumQExpOld = ewald.sumQExpNew.copy()
    s.s_ids = (ewald.sumQExpOld.copy(), ewald.sumQExpNew.copy())


def RecipRollback(ewald):
    """`ewald.sumQExpNew = [item for item in ewald.sumQExpOld]` (Ewald/main.jl:628)."""
    s = ewald._session
    s.ctx.recip_rollback()
    ewald.sumQExpNew = ewald.sumQExpOld.copy()
    s.s_ids = (ewald.sumQExpOld.copy(), ewald.sumQExpNew.copy())


def EwaldSelf(ewald, qq_q):
    """Ewald/ewalds.jl:829-833 (factor applied)."""
    s = getattr(ewald, "_session", None)
    if s is None or s.ctx.n_atoms != len(qq_q) or s.ctx._h is None:
        n = len(qq_q)
        s = _session_for_atoms(np.zeros((n, 3)), qq_q, 1.0 if s is None else s.box, ewald)
        s.bind_ewald(ewald, s.box)
    return s.ctx.ewald_self()


def potential(moa, soa, tot, ewalds, vdwTable, sim_props, coulomb_style=None):
    """potential(moa, soa, tot, ewalds, vdwTable, sim_props, "ewald")   Ewald/energy.jl:946-1032
    potential(moa, soa, tot, ewald, vdwTable, sim_props)   (Wolf)       Ewald/energy.jl:864-943
    -> tot::Properties (energy, virial, coulomb filled)."""
    box = sim_props.box
    s = _session(moa, soa, vdwTable, box)
    s.bind_ewald(ewalds, box)
    s.sync_all(moa, soa)
    if coulomb_style is None:
        t = s.ctx.potential_wolf(sim_props.LJ_rcut, sim_props.qq_rcut)
    else:
        t = s.ctx.potential_ewald(sim_props.LJ_rcut, sim_props.qq_rcut)
        s.pull_s(ewalds, old=True)  # RecipLong inside wrote both arrays (ewalds.jl:600-601)
    if tot is None:
        tot = Properties()
    tot.energy += t["energy"]
    tot.virial += t["virial"]
    tot.coulomb += t["coulomb"]
    tot.terms = t
    return tot
