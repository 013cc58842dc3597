import sys, os, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
import common
from metropolismontecarlo_amd import structs, io as mio
from metropolismontecarlo_amd.api import *
from metropolismontecarlo_amd.structs import EWALD, Properties, Properties2, Tables
a = common.nist_arrays(4, "unwrapped")
RCUT = 10.0
moa = structs.make_moa(a["com"].copy(), a["first_atom"], a["last_atom"])
soa = structs.make_soa(a["coords"].copy(), a["atype"], a["charge"])
vdwTable = Tables([mio.SPCE_EPS_O, 0.0], [mio.SPCE_SIGMA_O, 0.0])
box = a["box"]
ewald = EWALD(5.6 / box, 5, 27, 1, [[1, 1, 1]] * 3, [0.0, 0.0], np.zeros(2, complex), np.zeros(2, complex), structs.factor)
ewald = PrepareEwaldVariables(ewald, box)
totProps = Properties2(298.15, 0.0331, 0.0, 0.3166, 0.05, 0.3, 0, 0, [], RCUT, RCUT, box)
total = potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")
rng = np.random.default_rng(1)
def loop(n, timers=None):
    for s in range(n):
        i = s % 750 + 1
        f, l = moa.firstAtom[i-1], moa.lastAtom[i-1]
        t0 = time.perf_counter()
        e0, v0 = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
        t1 = time.perf_counter()
        q0, w0, o1 = EwaldShort(i, moa, soa, totProps, ewald, box)
        t2 = time.perf_counter()
        rm_old = moa.COM[i-1].copy(); ra_old = soa.coords[f-1:l].copy()
        d = (rng.random(3) - 0.5) * 0.3
        moa.COM[i-1] += d; soa.coords[f-1:l] += d
        ra_new = soa.coords[f-1:l].copy()
        t3 = time.perf_counter()
        e1, v1 = LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)
        q1, w1, o2 = EwaldShort(i, moa, soa, totProps, ewald, box)
        t4 = time.perf_counter()
        dr, _ = RecipMove(box, ewald, ra_old, ra_new, soa.charge[f-1:l])
        t5 = time.perf_counter()
        if s % 2:
            ewald.sumQExpOld = np.array([x for x in ewald.sumQExpNew])
        else:
            moa.COM[i-1] = rm_old; soa.coords[f-1:l] = ra_old
            ewald.sumQExpNew = np.array([x for x in ewald.sumQExpOld])
        if timers is not None:
            timers += np.array([t1-t0, t2-t1, t4-t3, t5-t4, time.perf_counter()-t0])
loop(50)
tm = np.zeros(5); n = 300
loop(n, tm)
import ctypes
from metropolismontecarlo_amd import api as _api
st = (ctypes.c_int64 * 10)()
sess = list(_api._sessions.values())[0]
sess._L.mmc_ctx_stats(sess._h, st)
print("ctx stats [cmds, launches, retries, cache hits, spec hits, spec miss, launch evals, alive]:", list(st))
print("api.py: us per LJ %.1f  EwaldShort %.1f  (LJ+ES new) %.1f  RecipMove %.1f  loop body %.1f" % tuple(1e6*tm/n))
# raw ctx
ctx = common.device_context(a)
ctx.recip_long()
def t(f, n=300):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return 1e6*(time.perf_counter()-t0)/n
ctx.lj_poly_du(7, RCUT)
us = ctypes.c_double()
for _ in range(2):
    ctx._L.mmc_ctx_ping(ctx._h, 2000, ctypes.byref(us))
print("server ping (empty command round trip): %.2f us" % us.value)
print("ctx: lj %.1f es %.1f setmol %.1f recip_move %.1f" % (t(lambda: ctx.lj_poly_du(5, RCUT)), t(lambda: ctx.ewald_short(5, RCUT)),
   t(lambda: ctx.set_molecule(5, a["com"][4], a["coords"][12:15])), t(lambda: ctx.recip_move(a["coords"][12:15], a["coords"][12:15]+0.1, a["charge"][12:15]))))
def tm_():
    d, o = ctx.trial_move(5, a["com"][4]+0.1, a["coords"][12:15]+0.1, RCUT, RCUT); ctx.reject_move()
print("ctx trial_move+reject %.1f" % t(tm_))
def tm2():
    d, o = ctx.trial_move(5, a["com"][4]+0.1, a["coords"][12:15]+0.1, RCUT, RCUT); ctx.accept_move()
print("ctx trial_move+accept %.1f" % t(tm2))
print("potential %.1f" % t(lambda: ctx.potential_ewald(RCUT,RCUT), 50))
