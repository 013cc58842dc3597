"""Build libmmc_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m metropolismontecarlo_amd.build [--force]

The library is built in-tree (metropolismontecarlo_amd/libmmc_hip.so) so that it travels with a
repository snapshot; it is git-ignored.  -ffp-contract=off keeps every branch decision of the pair
scan (gate, overlap, slack comparisons) bit-identical to an unfused evaluation, which is what the
reference's Julia does.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = ["mmc_hip.hip"]
DEPS = ["mmc_hip.hip", "mmc_wave_unit.inc", "mmc_host.hpp", "mmc_device.hpp", "mmc_kernels.hpp", "mmc_fast.hpp", "mmc_total.hpp", "mmc_wave.hpp",
        "mmc_propose.hpp", "mmc_study.hpp", "mmc_system.inc",
        "mmc_ctx.inc", "mmc_batch.inc", "mmc_engine.inc", "mmc_dist.inc", "mmc_ctxsrv.hpp",
        "mmc_lat.hpp", "mmc_wave_lat.inc", "mmc_potential.hpp"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(ROOT, "include", "mmc_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def command(extra=()):
    return [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
            "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-o", LIB,
            *[os.path.join(CSRC, s) for s in SOURCES], "-ldl", *extra]


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    cmd = command()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=ROOT)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
