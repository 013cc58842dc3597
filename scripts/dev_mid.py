"""Developer timing: the middle ground between the move server and the big launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
for R, persistent, wgs in ((128, 1, 2), (128, 0, -1), (256, 1, 0), (256, 0, -1), (512, 0, -1), (1024, 0, -1), (2048, 0, -1), (4096, 0, -1)):
    b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
              5.6 / a["box"], structs.factor, 10.0, 10.0)
    b.set_option("device_moves", 1)
    b.set_option("persistent", persistent)
    if wgs >= 0: b.set_option("server_wgs", wgs)
    e = b.potential_ewald(as_array=True)["energy"].copy()
    try:
        e, st = b.run(300, 298.15, 0.316555789, 0.05, 1, e, n_groups=2, n_threads=2 if R < 1024 else 4)
        n = 2000
        t0 = time.perf_counter()
        e, st = b.run(n, 298.15, 0.316555789, 0.05, 2, e, n_groups=2, n_threads=2 if R < 1024 else 4)
        dt = time.perf_counter() - t0
        print(f"R={R} persistent={persistent} wgs={wgs}: {1e6 * dt / n:.2f} us/step ({R * n / dt / 1e6:.2f} M moves/s) server_steps={st['server_steps']}")
    except Exception as ex:
        print(f"R={R} persistent={persistent}: {ex}")
    b.close()
