"""Developer timing of the move server shapes: us per step of R chains (750 molecules)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
import os
for R, wgs_list in eval(os.environ.get('LAT_CASES', '((1, (0, 2, 3, 4)), (2, (0, 2, 4)), (4, (0, 2, 4)), (32, (0, 2)))')):
    for wgs in wgs_list:
        b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
                  5.6 / a["box"], structs.factor, 10.0, 10.0)
        b.set_option("device_moves", 1)
        b.set_option("server_wgs", wgs)
        e = b.potential_ewald(as_array=True)["energy"].copy()
        e, st = b.run(600, 298.15, 0.316555789, 0.05, 1, e, n_groups=1 if R == 1 else 2, n_threads=1 if R == 1 else 2)
        t0 = time.perf_counter()
        n = 4000
        e, st = b.run(n, 298.15, 0.316555789, 0.05, 2, e, n_groups=1 if R == 1 else 2, n_threads=1 if R == 1 else 2)
        dt = time.perf_counter() - t0
        e2 = b.potential_ewald(as_array=True)["energy"]
        print(f"R={R} server_wgs={wgs}: {1e6 * dt / n:.2f} us/step  server_steps={st['server_steps']} drift={np.abs(e - e2).max() / np.abs(e2).max():.1e}")
        b.close()
