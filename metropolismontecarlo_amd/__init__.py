"""metropolismontecarlo_amd -- MI355X (gfx950) implementation of the per-move energy hot path of
BradenDKelly/MetropolisMonteCarlo behind the reference's own call surface.

    api      LJ_poly_ΔU, EwaldReal, EwaldShort, RecipLong, RecipMove, PrepareEwaldVariables,
             EwaldSelf, potential, CoulombReal  (reference names and signatures)
    device   Context / Batch: object wrappers over the C ABI (include/mmc_hip.h)
    structs  EWALD, Tables, Properties, Properties2, Requirements, moa/soa columns
    io       ReadNIST and the synthetic lattice used by the benchmarks
    build    hipcc build of libmmc_hip.so

The compute path is hand-written HIP in csrc/; there is no CPU fallback.
"""
from .structs import (EWALD, Moves, Properties, Properties2, Requirements, StructArray, Tables,
                      factor, make_moa, make_soa)

__version__ = "0.1.0"
