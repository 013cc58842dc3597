"""Developer timing: chains on the move server, workgroups per replica x host threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
cases = eval(os.environ.get("CASES", "[(32,4,4),(32,2,4),(64,4,4),(64,3,4),(64,2,4),(128,2,4),(128,2,8),(64,4,8),(16,4,2),(16,4,4)]"))
for R, wgs, threads in cases:
    b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
              5.6 / a["box"], structs.factor, 10.0, 10.0)
    b.set_option("device_moves", 1)
    b.set_option("server_wgs", wgs)
    e = b.potential_ewald(as_array=True)["energy"].copy()
    e, st = b.run(600, 298.15, 0.316555789, 0.05, 1, e, n_groups=2, n_threads=threads)
    n = 4000
    t0 = time.perf_counter()
    e, st = b.run(n, 298.15, 0.316555789, 0.05, 2, e, n_groups=2, n_threads=threads)
    dt = time.perf_counter() - t0
    print(f"R={R} wgs={wgs} threads={threads}: {1e6 * dt / n:.2f} us/step ({R * n / dt / 1e6:.2f} M moves/s)")
    b.close()
