#!/usr/bin/env python3
"""M2 (ns per full energy evaluation): batched over R replicas and as single-system latency.
  python3 scripts/m2_bench.py [R]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: F401
from metropolismontecarlo_amd import io as mio
from metropolismontecarlo_amd import structs, io as mio
from metropolismontecarlo_amd.device import Batch, Context
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
a = mio.load_nist_fixture(4, "unwrapped")
b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
          5.6 / a["box"], structs.factor, 10.0, 10.0)
b.potential_ewald(as_array=True)
n = 5
t0 = time.perf_counter()
for _ in range(n):
    t = b.potential_ewald(as_array=True)
dt = (time.perf_counter() - t0) / n
print(f"batched R={R}: {1e9*dt/R:.1f} ns per evaluation ({1e3*dt:.2f} ms per call), E={t['energy'][0]:.6f}")
b.close()
for nm in (750, 10000):
    if nm == 750:
        s = a
        first = a["first_atom"]; last = a["last_atom"]
    else:
        box, com, coords = mio.cubic_lattice_water(nm, 0.033101144, "spce", seed=11234)
        first = 3 * np.arange(nm, dtype=np.int64) + 1; last = first + 2
        s = dict(com=com, coords=coords, atype=np.tile([1, 2, 2], nm), charge=np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], nm),
                 eps=a["eps"], sig=a["sig"], box=box)
    ctx = Context()
    ctx.upload_system(s["com"], first, last, s["coords"], s["atype"], s["charge"], s["eps"], s["sig"], s["box"])
    ctx.prepare_ewald(5.6 / s["box"], 5, 27, s["box"], structs.factor)
    ctx.potential_ewald(10.0, 10.0)
    n = 50 if nm == 750 else 20
    t0 = time.perf_counter()
    for _ in range(n):
        e = ctx.potential_ewald(10.0, 10.0)
    dt = (time.perf_counter() - t0) / n
    print(f"single system {nm} molecules: {1e6*dt:.1f} us per potential(), E={e['energy']:.6f}")
    ctx.close()
