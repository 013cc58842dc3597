"""Developer timing: 256 chains, server (one workgroup per replica) against launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
for R, persistent, threads in ((256, 1, 8), (256, 1, 4), (256, 0, 4), (256, 0, 8), (192, 1, 8), (192, 0, 4)):
    b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
              5.6 / a["box"], structs.factor, 10.0, 10.0)
    b.set_option("device_moves", 1)
    b.set_option("persistent", persistent)
    e = b.potential_ewald(as_array=True)["energy"].copy()
    e, st = b.run(300, 298.15, 0.316555789, 0.05, 1, e, n_groups=2, n_threads=threads)
    n = 2000
    t0 = time.perf_counter()
    e, st = b.run(n, 298.15, 0.316555789, 0.05, 2, e, n_groups=2, n_threads=threads)
    dt = time.perf_counter() - t0
    print(f"R={R} persistent={persistent} threads={threads}: {1e6 * dt / n:.2f} us/step ({R * n / dt / 1e6:.2f} M moves/s) server_steps={st['server_steps']}")
    b.close()
