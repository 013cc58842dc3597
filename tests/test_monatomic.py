"""BASELINE configs[0]: the monatomic Lennard-Jones NVT driver (Monatomic/mainMonatomic.jl:345-413)
as a workload -- 256 atoms, rho* = 0.75, T* = 1, r_c = 2.5, dr_max = box / 30 -- run through the
CPU oracle's `LJ_dU` (oracle/mmc_oracle.c, pinned on the reference's own `test_LJ` in
test_oracle.py).  SURVEY.md section 8 row a14 keeps this path on the CPU ("plumbing, no GPU").

Reference-held pins used here: the acceptance ratio the author notes for exactly these
parameters, "at 256 particles, rho=0.75, T=1.0 this is 48% acceptance" (mainMonatomic.jl:351), and
the running-total-equals-recompute invariant (Poly/main.jl:232-235).  Julia's RNG stream is not
reproducible here (MersenneTwister seeded 11234, no Julia in the image), so the chain draws from
numpy with the same seed: the acceptance is a statistical pin, not a bitwise one."""
import math

import numpy as np

from metropolismontecarlo_amd import io as mio
from metropolismontecarlo_amd import moves
from oracle import oracle as orc

N_ATOMS, RHO, TEMPERATURE, R_CUT, SEED = 256, 0.75, 1.0, 2.5, 11234   # mainMonatomic.jl:15,38-44


def test_init_cubic_grid_matches_reference_layout():
    L, r = mio.InitCubicGrid(N_ATOMS, RHO)
    assert math.isclose(L, (N_ATOMS / RHO) ** (1 / 3)) and math.isclose(L, 6.9886, abs_tol=1e-4)
    d = L / 7                                   # 7^3 = 343 is the smallest cube >= 256
    assert np.allclose(r[0], 0.01 * d) and np.allclose(r[1], [1.01 * d, 0.01 * d, 0.01 * d])
    assert np.allclose(r[7], [0.01 * d, 1.01 * d, 0.01 * d])       # x runs fastest, then y
    assert np.allclose(r[49], [0.01 * d, 0.01 * d, 1.01 * d])
    assert r.shape == (N_ATOMS, 3) and r.min() > 0 and r.max() < L
    L2, r2 = mio.InitCubicGrid(8, 1.0)          # nCube starts at 2
    assert math.isclose(L2, 2.0) and np.allclose(r2[-1], [1.01, 1.01, 1.01])


def test_monatomic_nvt_chain_acceptance_and_running_total():
    box, r = mio.InitCubicGrid(N_ATOMS, RHO)
    r = np.ascontiguousarray(r)
    eps, sig = np.ones(N_ATOMS), np.ones(N_ATOMS)
    dr_max = box / 30                                            # mainMonatomic.jl:350
    rng = np.random.default_rng(SEED)
    e0, v0 = orc.potential_monatomic(r, eps, sig, R_CUT, box)    # :365
    total_e, total_v = e0, v0
    n_equil, n_meas = 60, 200                                    # sweeps
    acc = att = 0
    for sweep in range(n_equil + n_meas):
        for i in range(1, N_ATOMS + 1):                          # :375
            old_e, old_v = orc.lj_du_monatomic(i, r, eps, sig, R_CUT, box)       # :377
            rold = r[i - 1].copy()
            r[i - 1] = moves.random_translate_vector(dr_max, rold, box, rng)     # :379
            new_e, new_v = orc.lj_du_monatomic(i, r, eps, sig, R_CUT, box)       # :381
            delta = new_e - old_e
            ok = moves.Metropolis(delta / TEMPERATURE, rng)                      # :385
            if ok:
                total_e += delta
                total_v += new_v - old_v
            else:
                r[i - 1] = rold
            if sweep >= n_equil:
                att += 1
                acc += ok
        assert r.min() >= 0.0 and r.max() <= box                 # the driver's in-box check :399-405
    ratio = acc / att
    assert 0.44 < ratio < 0.52, ratio                            # "48% acceptance" (:351)
    e1, v1 = orc.potential_monatomic(r, eps, sig, R_CUT, box)
    assert abs(total_e - e1) < 1e-9 * abs(e1), (total_e, e1)     # Poly/main.jl:232-235
    assert abs(total_v - v1) < 1e-9 * abs(v1), (total_v, v1)
    # the fluid has melted off the lattice: energy per atom near the literature value for the cut
    # (not shifted) potential at this state point, about -4.3 (without the tail correction)
    assert -5.0 < e1 / N_ATOMS < -3.8
