"""Developer timing: trial moves of ONE 10 000-molecule SPC/E system (BASELINE configs[3] without
the volume moves) and of a few such chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Batch
nm = 10000
box, com, coords = mio.cubic_lattice_water(nm, 0.033101144, "spce", seed=11234)
a4 = mio.load_nist_fixture(4, "unwrapped")
atype = np.tile([1, 2, 2], nm); charge = np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], nm)
for R in (1, 8, 64):
    b = Batch(R, com, coords, atype, charge, a4["eps"], a4["sig"], box, 5.6 / box, structs.factor, 10.0, 10.0)
    b.set_option("device_moves", 1)
    e = b.potential_ewald(as_array=True)["energy"].copy()
    e, st = b.run(200, 298.15, 0.3, 0.05, 1, e, n_groups=min(R, 2), n_threads=2)
    n = 2000
    t0 = time.perf_counter()
    e, st = b.run(n, 298.15, 0.3, 0.05, 2, e, n_groups=min(R, 2), n_threads=2)
    dt = time.perf_counter() - t0
    e2 = b.potential_ewald(as_array=True)["energy"]
    print(f"10000 molecules, R={R}: {1e6*dt/n:.2f} us per step, server_steps={st['server_steps']}, launches={st['launches']}, accept {(st['trans_accept']+st['rot_accept'])/st['moves']:.2f}, drift {np.abs(e-e2).max()/np.abs(e2).max():.1e}")
    b.close()
