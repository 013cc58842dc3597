#!/usr/bin/env python3
"""Persistent move server against the launch-per-step driver on the same units: identical chains."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: F401
from metropolismontecarlo_amd import io as mio
from metropolismontecarlo_amd import structs
from metropolismontecarlo_amd.device import Batch
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
a = mio.load_nist_fixture(k, "unwrapped")
n_mol = a["com"].shape[0]
P = int(os.environ.get('SRV_P', min((n_mol + 63) // 64, 7) + 1))
big = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for R, steps in (((1, big), (32, big)) if big else ((1, 40), (3, 2 * n_mol + 5), (32, 300))):
    res = []
    for persistent in (1, 0):
        b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
                  5.6 / a["box"], structs.factor, 10.0, 10.0)
        b.set_option("device_moves", 1)
        b.set_option("kernel", 2)
        b.set_option("persistent", persistent)
        e0 = b.potential_ewald(as_array=True)["energy"].copy()
        if big:  # warm: allocations, first launch
            e0, _ = b.run(50, 298.15, 0.316555789, 0.05, 7, e0, n_groups=1 if R == 1 else 2, n_parts=P,
                          n_threads=1 if R == 1 else 2)
        t0 = time.perf_counter()
        e1, st = b.run(steps, 298.15, 0.316555789, 0.05, 7, e0, n_groups=1 if R == 1 else 2, n_parts=P,
                       n_threads=1 if R == 1 else 2)
        dt = time.perf_counter() - t0
        e2 = b.potential_ewald(as_array=True)["energy"]
        com, coords, S = b.get_replica(R - 1)
        res.append((e1, st, com, coords, S))
        print(f"R={R} steps={steps} persistent={persistent}: {1e6*dt/steps:.2f} us/step, accept "
              f"{(st['trans_accept']+st['rot_accept'])/st['moves']:.3f}, drift {np.abs(e1-e2).max()/np.abs(e2).max():.1e}, torn {st['torn_records']}", flush=True)
        b.close()
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][2], res[1][2]) and \
        np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])
    print("   identical chains:", same, flush=True)
    assert same or os.environ.get('SRV_NOASSERT')
print("ok")
