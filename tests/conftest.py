import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _gpu_selected(config):
    expr = (config.getoption("-m") or "").replace(" ", "")
    return "gpu" in expr and "notgpu" not in expr


def _device_present():
    # /dev/kfd is the ROCm compute device node; looking at it initialises nothing
    return os.path.exists("/dev/kfd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config._mmc_dist = None
    if _gpu_selected(config) and _device_present():
        # The N>1 rehearsal of the product path (tests/test_gpu_dist.py) runs as child processes
        # that are started HERE, before anything in this process has touched the GPU: a process
        # that has initialised HIP must not start other programs on this pool.  The children run
        # beside the other tests and the test only collects their output.
        out = tempfile.mkdtemp(prefix="mmc_dist_")
        env = dict(os.environ, MMC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
        common = ["--steps", "20", "--warmup", "5", "--no-cpu", "--no-secondary", "--threads", "1"]
        script = (
            f"cd {ROOT} && "
            f"{sys.executable} -m torch.distributed.run --nnodes=1 --nproc-per-node 2 "
            f"--master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --replicas 64 "
            f"{' '.join(common)} > {out}/two.json 2> {out}/two.err; echo $? > {out}/two.rc; "
            f"{sys.executable} bench.py --gpus 1 --replicas 128 {' '.join(common)} "
            f"> {out}/one.json 2> {out}/one.err; echo $? > {out}/one.rc; "
            # BASELINE configs[2]'s share of a GPU (32 chains) under two ranks: the move server
            f"{sys.executable} -m torch.distributed.run --nnodes=1 --nproc-per-node 2 "
            f"--master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --replicas 32 "
            f"--steps 200 --warmup 20 --no-cpu --no-secondary --threads 2 "
            f"> {out}/two32.json 2> {out}/two32.err; echo $? > {out}/two32.rc; "
            # the headline's own mode under two ranks: the move kernel decides, eight steps per launch
            f"{sys.executable} -m torch.distributed.run --nnodes=1 --nproc-per-node 2 "
            f"--master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 2 --replicas 10240 "
            f"--steps 24 --warmup 8 --no-cpu --no-secondary --threads 2 "
            f"> {out}/two_dev.json 2> {out}/two_dev.err; echo $? > {out}/two_dev.rc; "
            f"{sys.executable} bench.py --gpus 1 --replicas 20480 --steps 24 --warmup 8 --no-cpu "
            f"--no-secondary --threads 2 > {out}/one_dev.json 2> {out}/one_dev.err; echo $? > {out}/one_dev.rc")
        proc = subprocess.Popen(["bash", "-c", script], env=env)
        config._mmc_dist = (proc, out)


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a machine without a device must fail loudly, not pass by deselecting or
    # skipping: a silent skip would read as "parity green".
    if _gpu_selected(config) and not _device_present():
        raise pytest.UsageError("-m gpu selected but no ROCm device (/dev/kfd) on this machine: "
                                "the gpu tests need a real MI355X (use gpurun)")


@pytest.fixture(scope="session")
def dist_rehearsal(request):
    """(directory with two.json / one.json / *.err / *.rc) once the child launched at start-up has
    finished."""
    h = request.config._mmc_dist
    if h is None:
        pytest.skip("the N>1 rehearsal is launched only under -m gpu on a machine with a device")
    proc, out = h
    proc.wait(timeout=600)
    return out


def pytest_unconfigure(config):
    h = getattr(config, "_mmc_dist", None)
    if h is not None and h[0].poll() is None:
        h[0].wait(timeout=600)
