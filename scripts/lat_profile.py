"""Phase timing of k_move_server_lat from a -DLAT_PROFILE build (MMC_HIP_LIB=<that build>): one chain,
4000 steps; per wave of replica 0 the mean time per step spent waiting for the control word, in the
commit, in the unit body, at the barrier, storing the record, making the next proposal."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs, _lib
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
          5.6 / a["box"], structs.factor, 10.0, 10.0)
b.set_option("device_moves", 1)
e = b.potential_ewald(as_array=True)["energy"].copy()
e, st = b.run(500, 298.15, 0.316555789, 0.05, 1, e, n_groups=1, n_threads=1)
n = 4000
t0 = time.perf_counter()
e, st = b.run(n, 298.15, 0.316555789, 0.05, 2, e, n_groups=1, n_threads=1)
dt = time.perf_counter() - t0
print(f"{1e6 * dt / n:.2f} us per step, server_steps {st['server_steps']}")
L = _lib.lib()
buf = np.zeros((64, 8), dtype=np.uint64)
L.mmc_debug_lat_profile.argtypes = [C.c_void_p]
assert L.mmc_debug_lat_profile(buf.ctypes.data_as(C.c_void_p)) == 0
names = ["wait word", "commit", "unit", "barrier", "record", "proposal"]
print("wave " + " ".join(f"{x:>10s}" for x in names) + "      sum (us per step)")
for w in range(64):
    if buf[w].sum() == 0: continue
    v = buf[w, :6].astype(float) * 0.01 / n
    setup_scan = float(int(buf[w, 6]) & 0xffffffff) * 0.01 / n
    rounds = float(int(buf[w, 6]) >> 32) * 0.01 / n
    sums = float(buf[w, 7]) * 0.01 / n
    print(f"{w:4d} " + " ".join(f"{x:10.2f}" for x in v) + f" {v.sum():10.2f}   unit: setup+scan {setup_scan:.2f} rounds {rounds:.2f} sums {sums:.2f}")
b.close()
