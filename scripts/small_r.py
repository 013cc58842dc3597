#!/usr/bin/env python3
"""Latency of the per-move path at BASELINE's named replica counts (configs[1]: one chain on one
GPU, configs[2]: 256 chains over 8 GPUs = 32 per GPU): us per step for a few launch shapes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: F401
from metropolismontecarlo_amd import io as mio
from metropolismontecarlo_amd import structs
from metropolismontecarlo_amd.device import Batch
a = mio.load_nist_fixture(4, "unwrapped")
for R in (1, 32):
    for kernel in (2, 1):
        for dev in (1, 0):
            for parts in (1, 3, 5, 9, 16):
                b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
                          5.6 / a["box"], structs.factor, 10.0, 10.0)
                b.set_option("kernel", kernel); b.set_option("device_moves", dev)
                b.set_option("zero_copy_moves", 1)
                e = b.potential_ewald(as_array=True)["energy"].copy()
                g = 1 if R == 1 else 2
                e, _ = b.run(300, 298.15, 0.316555789, 0.05, 1, e, n_groups=g, n_parts=parts, n_threads=g)
                t0 = time.perf_counter()
                e, st = b.run(3000, 298.15, 0.316555789, 0.05, 2, e, n_groups=g, n_parts=parts, n_threads=g)
                dt = time.perf_counter() - t0
                print(f"R={R:3d} kernel={kernel} device_moves={dev} parts={parts:2d}: {1e6*dt/3000:7.2f} us/step  {R*3000/dt:10.0f} moves/s", flush=True)
                b.close()
