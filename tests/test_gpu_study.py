"""The single-precision study kernels (csrc/mmc_study.hpp; BASELINE configs[4]) against the fp64
product path on a NIST configuration: same terms, single-precision error levels."""
import numpy as np
import pytest

import common
from common import rel

pytestmark = pytest.mark.gpu

RCUT = 10.0


def test_study_totals_and_moves_track_fp64():
    a = common.nist_arrays(2, "unwrapped")
    with common.device_context(a) as ctx:
        e64 = ctx.potential_ewald(RCUT, RCUT)
        for mixed, tol in ((False, 2e-5), (True, 5e-6)):
            t = ctx.study_f32_total(RCUT, RCUT, mixed)
            assert t["n_overlap"] == e64["n_overlap"] == 0
            assert rel(t["lj"], e64["lj"]) < tol
            assert rel(t["real"], e64["real"]) < tol
            assert rel(t["recip"], e64["recip"]) < 50 * tol      # |S|^2 of O(1e2) charges in fp32
        rng = np.random.default_rng(5)
        n_mol, box = a["com"].shape[0], a["box"]
        worst = 0.0
        for _ in range(40):
            i = int(rng.integers(1, n_mol + 1))
            d = (rng.random(3) - 0.5) * 0.3
            c_new = a["com"][i - 1] + d
            a_new = a["coords"][3 * (i - 1):3 * i] + d
            d64, ov = ctx.trial_move(i, c_new, a_new, RCUT, RCUT)
            ctx.reject_move()
            for mixed in (False, True):
                d32, ov32 = ctx.study_f32_move(i, c_new, a_new, RCUT, RCUT, mixed)
                assert ov32 == ov
                # fp32 error of a difference of O(1e4 K) sums: well below 1 K, far above 1e-6 rel
                assert np.abs(d32 - d64[:3]).max() < 1.0
                worst = max(worst, np.abs(d32 - d64[:3]).max())
        assert worst > 1e-6      # it really is single precision
        assert box == a["box"]


def test_study_needs_total_first_and_valid_arguments():
    from metropolismontecarlo_amd._lib import MMCError
    a = common.nist_arrays(1, "unwrapped")
    with common.device_context(a) as ctx:
        with pytest.raises(MMCError, match="MMC_ERR_STATE"):
            ctx.study_f32_move(1, a["com"][0], a["coords"][:3], RCUT, RCUT)
        ctx.potential_ewald(RCUT, RCUT)
        ctx.study_f32_total(RCUT, RCUT)
        with pytest.raises(MMCError, match="MMC_ERR_ARG"):
            ctx.study_f32_move(0, a["com"][0], a["coords"][:3], RCUT, RCUT)
        d, ov = ctx.study_f32_move(1, a["com"][0], a["coords"][:3], RCUT, RCUT)   # null move
        assert not ov and np.abs(d).max() < 1e-2
