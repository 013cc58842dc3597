"""Accuracy of the fast kernel's erfc(kappa r)/r approximation (mmc_fast.hpp) against an exact
(mpmath, 40 digits) evaluation of the reference's expression `erfc(kappa * rab_mag) / rab_mag`
(Ewald/ewalds.jl:365-367), over the whole range of r^2 the kernel can meet.

Bound asserted: 4e-14 relative over the whole domain (reached only where kappa*r > 3, i.e. where
the term itself is below 1e-5 of its neighbours'), 5e-15 for kappa*r <= 2.5.  The reference's own
fp64 evaluation of that expression is off by up to ~5e-15 at kappa*r ~ 4 because erfc amplifies
the rounding of its argument by 2x^2; the table is built from such fp64 samples (degree 9 per
piece, 16 pieces per octave)."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def exact(kappa, u):
    import mpmath as mp
    mp.mp.dps = 40
    return np.array([float(mp.erfc(kappa * mp.sqrt(mp.mpf(float(x)))) / mp.sqrt(mp.mpf(float(x))))
                     for x in u])


@pytest.mark.parametrize("box", [20.0, 30.0, 67.099, 200.0])
def test_qq_table_accuracy(box):
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Batch
    a = common.nist_arrays(1)
    kappa = 5.6 / box
    rng = np.random.default_rng(int(box))
    # the kernel is selected only when kappa*sqrt(r_cut^2 + 100) <= 4 and r_cut^2 + 100 <= 256
    u_hi = min(255.999, 16.0 / kappa ** 2)
    u = np.concatenate([
        np.exp(rng.uniform(np.log(0.25), np.log(u_hi), 6000)),     # the table proper
        np.exp(rng.uniform(np.log(1e-4), np.log(0.25), 500)),      # small-r series branch
        [0.25, 0.5, 1.0, 2.0 - 1e-15, 200.0, u_hi],
        2.0 ** np.arange(-2, 8) * (1 + 1.0 / 16),                  # piece boundaries
        np.nextafter(2.0 ** np.arange(-1, 8), 0),
    ])
    u = u[u <= u_hi]
    # the table depends on kappa only; the system just has to exist (box must match the Ewald box)
    b = Batch(1, a["com"] % box, a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], box,
              kappa, structs.factor, 9.0, 9.0)
    try:
        got = b.qq_table(u)
    finally:
        b.close()
    ref = exact(kappa, u)
    err = np.abs(got - ref) / ref
    assert err.max() < 4e-14, (box, u[err.argmax()], err.max())
    near = kappa * np.sqrt(u) <= 2.5
    assert err[near].max() < 5e-15, (box, u[near][err[near].argmax()], err[near].max())
    # in absolute terms (what a sum of such terms sees) it is as good as the direct fp64 formula
    # the oracle/reference use
    from scipy.special import erfc
    direct = erfc(kappa * np.sqrt(u)) / np.sqrt(u)
    abs_direct = np.abs(direct - ref) * np.sqrt(u)          # relative to the bare 1/r term
    abs_table = np.abs(got - ref) * np.sqrt(u)
    assert abs_table.max() < 4 * max(abs_direct.max(), 1e-15)


def test_qq_table_domain_errors():
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd._lib import MMCError
    from metropolismontecarlo_amd.device import Batch
    a = common.nist_arrays(1)
    with Batch(1, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
               5.6 / a["box"], structs.factor, 10.0, 10.0) as b:
        with pytest.raises(MMCError, match="MMC_ERR_ARG"):
            b.qq_table([256.0])
        with pytest.raises(MMCError, match="MMC_ERR_ARG"):
            b.qq_table([0.0])
        assert b.qq_table([]).shape == (0,)
