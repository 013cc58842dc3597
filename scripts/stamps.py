#!/usr/bin/env python3
"""Phase timing of k_move_eval_wave from a -DWV_STAMPS build (MMC_HIP_LIB=<that build>):
runs a few steps of the default bench workload and prints, per phase, the mean shader-clock
cycles a wave spent in it (lane 0 stamps; each stamp drains the wave's memory counters first)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: F401
from metropolismontecarlo_amd import io as mio
from metropolismontecarlo_amd import structs, _lib
from metropolismontecarlo_amd.device import Batch
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
a = mio.load_nist_fixture(4, "unwrapped")
b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
          5.6 / a["box"], structs.factor, 10.0, 10.0)
b.set_option("device_moves", 1)
e = b.potential_ewald(as_array=True)["energy"].copy()
e, st = b.run(12, 298.15, 0.316555789, 0.05, 11234, e, n_groups=1, n_threads=2)
L = _lib.lib()
n = min(R, 65536)
buf = np.zeros((n, 8), dtype=np.uint64)
L.mmc_debug_stamps.argtypes = [C.c_void_p, C.c_int64]
assert L.mmc_debug_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
t = buf.astype(np.int64)
names = ["record load", "phase tables", "recip loop", "scan", "first gather", "pair loops", "reduce+store"]
d = np.diff(t, axis=1)
ok = (d >= 0).all(axis=1) & (d < 10_000_000).all(axis=1)
d = d[ok]
print(f"units {ok.sum()} of {n}; total per unit mean {d.sum(axis=1).mean():.0f} shader cycles")
for k, nm in enumerate(names):
    print(f"  {nm:14s} mean {d[:, k].mean():9.1f}  median {np.median(d[:, k]):9.1f}  p90 {np.percentile(d[:, k], 90):9.1f}")
b.close()
