"""Developer timing: latency of one potential(..., "ewald") / RecipLong call of ONE system."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Context
a = mio.load_nist_fixture(4, "unwrapped")
def mk(nm):
    if nm == 750:
        return a, a["first_atom"], a["last_atom"]
    box4, com4, coords4 = mio.cubic_lattice_water(nm, 0.033101144, "spce", seed=11234)
    first = 3 * np.arange(nm, dtype=np.int64) + 1
    return dict(com=com4, coords=coords4, atype=np.tile([1, 2, 2], nm), charge=np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], nm), eps=a["eps"], sig=a["sig"], box=box4), first, first + 2
for nm in (750, 10000):
    s, first, last = mk(nm)
    ctx = Context()
    ctx.upload_system(s["com"], first, last, s["coords"], s["atype"], s["charge"], s["eps"], s["sig"], s["box"])
    ctx.prepare_ewald(5.6 / s["box"], 5, 27, s["box"], structs.factor)
    e = ctx.potential_ewald(10.0, 10.0)
    n = 200 if nm == 750 else 50
    t0 = time.perf_counter()
    for _ in range(n):
        e = ctx.potential_ewald(10.0, 10.0)
    t1 = time.perf_counter()
    for _ in range(n):
        r = ctx.recip_long()
    t2 = time.perf_counter()
    print(f"{nm} molecules: potential {1e6 * (t1 - t0) / n:.1f} us  RecipLong {1e6 * (t2 - t1) / n:.1f} us  E = {e['energy']:.6f} recip = {r:.9f}")
    ctx.close()
