"""Developer check + timing: mmc_trial_move on a 10 000-molecule context through the context server,
against the same moves with a launch per evaluation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metropolismontecarlo_amd import io as mio, structs
from metropolismontecarlo_amd.device import Context
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
box, com, coords = mio.cubic_lattice_water(nm, 0.033101144, "spce", seed=11234)
a4 = mio.load_nist_fixture(4, "unwrapped")
first = 3 * np.arange(nm, dtype=np.int64) + 1
res = {}
for server in (1, 0):
    ctx = Context()
    ctx.upload_system(com, first, first + 2, coords, np.tile([1, 2, 2], nm),
                      np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], nm), a4["eps"], a4["sig"], box)
    ctx.prepare_ewald(5.6 / box, 5, 27, box, structs.factor)
    ctx.set_option("server", -1 if server else 0)
    ctx.recip_long()
    rng = np.random.default_rng(3)
    c, x = com.copy(), coords.copy()
    out = []
    n = 400
    t0 = time.perf_counter()
    for k in range(n):
        i = k % nm + 1
        d = (rng.random(3) - 0.5) * 0.3
        cn, an = c[i - 1] + d, x[3 * (i - 1):3 * i] + d
        du, ov = ctx.trial_move(i, cn, an, 10.0, 10.0)
        out.append(du.copy())
        if k % 3 and not ov:
            ctx.accept_move(); c[i - 1] = cn; x[3 * (i - 1):3 * i] = an
        else:
            ctx.reject_move()
    dt = time.perf_counter() - t0
    print(f"{nm} molecules, server={server}: {1e6 * dt / n:.1f} us per trial move; stats {ctx.stats() if server else ''}")
    res[server] = np.array(out)
    ctx.close()
err = np.abs(res[1] - res[0]).max() / (np.abs(res[0]).max() + 1e4)
print("max relative difference server vs launch-per-evaluation:", err)
assert err < 1e-12
