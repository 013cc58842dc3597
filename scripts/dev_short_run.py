"""Developer timing: where a 20-step mmc_batch_run call spends its time beyond 40 kernel launches."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
R = 65536
b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"], 5.6 / a["box"], structs.factor, 10.0, 10.0)
b.set_option("device_moves", 1)
e = b.potential_ewald(as_array=True)["energy"].copy()
e, st = b.run(150, 298.15, 0.316555789, 0.05, 1, e, n_groups=2, n_threads=8, time_kernels=8)
for n in (1, 2, 5, 10, 20, 20, 20, 40, 100, 300):
    t0 = time.perf_counter()
    e, st = b.run(n, 298.15, 0.316555789, 0.05, 2, e, n_groups=2, n_threads=8, time_kernels=8)
    dt = time.perf_counter() - t0
    k = st["kernel_ms"] / max(st["timed_launches"], 1)
    print(f"n={n}: python {1e3*dt:.3f} ms, C wall {st['wall_ms']:.3f} ms, per step {1e3*dt/n:.4f} ms, kernel {1e3*k:.1f} us, ideal {2*n*k:.3f} ms, excess {1e3*dt - 2*n*k:.3f} ms")
b.close()
