"""Soak: long runs on the move server (and one with launches) -- no torn record may slip through, no
wait may time out, the running totals must still equal a recompute."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
for R, n, threads in ((1, 300000, 1), (32, 150000, 4), (256, 40000, 8), (4096, 6000, 4)):
    b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
              5.6 / a["box"], structs.factor, 10.0, 10.0)
    b.set_option("device_moves", 1)
    e = b.potential_ewald(as_array=True)["energy"].copy()
    t0 = time.perf_counter()
    e, st = b.run(n, 298.15, 0.316555789, 0.05, 5, e, n_groups=2 if R > 1 else 1, n_threads=threads)
    dt = time.perf_counter() - t0
    e2 = b.potential_ewald(as_array=True)["energy"]
    drift = np.abs(e - e2).max() / np.abs(e2).max()
    print(f"R={R}: {n} steps in {dt:.2f} s ({1e6*dt/n:.2f} us/step), server_steps={st['server_steps']}, torn={st['torn_records']}, "
          f"accept {(st['trans_accept']+st['rot_accept'])/st['moves']:.3f}, drift {drift:.1e}", flush=True)
    assert drift < 1e-10
    b.close()
