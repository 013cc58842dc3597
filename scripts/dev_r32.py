"""Developer timing: 32 chains on the move server under different driver shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
R = 32
for groups, threads, wgs in ((2, 2, 2), (2, 1, 2), (4, 4, 2), (8, 8, 2), (1, 1, 2), (4, 4, 4), (8, 8, 4), (8, 4, 2)):
    b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
              5.6 / a["box"], structs.factor, 10.0, 10.0)
    b.set_option("device_moves", 1)
    b.set_option("server_wgs", wgs)
    e = b.potential_ewald(as_array=True)["energy"].copy()
    e, st = b.run(600, 298.15, 0.316555789, 0.05, 1, e, n_groups=groups, n_threads=threads)
    n = 4000
    t0 = time.perf_counter()
    e, st = b.run(n, 298.15, 0.316555789, 0.05, 2, e, n_groups=groups, n_threads=threads)
    dt = time.perf_counter() - t0
    print(f"R=32 groups={groups} threads={threads} wgs={wgs}: {1e6 * dt / n:.2f} us/step ({R * n / dt / 1e6:.2f} M moves/s)")
    b.close()
