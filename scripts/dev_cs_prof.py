"""Developer timing of the context server's phases (build with -DCS_PROFILE, MMC_HIP_LIB=...)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Context
a = mio.load_nist_fixture(4, "unwrapped")
ctx = Context()
ctx.upload_system(a["com"], a["first_atom"], a["last_atom"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"])
ctx.prepare_ewald(5.6 / a["box"], 5, 27, a["box"], structs.factor)
ctx.recip_long()
rng = np.random.default_rng(1)
two = len(sys.argv) > 1 and sys.argv[1] == "two"
n = 3000
t0 = time.perf_counter()
for s in range(n):
    i = s % 750 + 1
    d = (rng.random(3) - 0.5) * 0.3
    if two:
        dd, ov = ctx.trial_move(i, a["com"][i - 1] + d, a["coords"][3 * i - 3:3 * i] + d, 10.0, 10.0)
        ctx.reject_move()
    else:
        ctx.set_molecule(i, a["com"][i - 1] + d, a["coords"][3 * i - 3:3 * i] + d)
        ctx.lj_poly_du(i, 10.0)
print("us per call (python):", 1e6 * (time.perf_counter() - t0) / n, ctx.stats())
ctx.close()
