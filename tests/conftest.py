import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")


def bucket_xy(W, b):
    """Bucket column and row of bucket b, restated from include/frayhip.h: by = b / BW, bx = (b % BW + FRAYHIP_BUCKET_SKEW * by) % BW
    with FRAYHIP_BUCKET_SKEW = 3 (tests/test_abi.py checks this restatement against frayhip_bucket_xy)."""
    BW = (W - 1) // 48 + 1
    by = b // BW
    return (b % BW + 3 * by) % BW, by


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # A clean launcher for tests that start other programs (torch.distributed.run + bench.py): the fork server is started now,
    # before anything in this process has touched the GPU, and children are forked from IT -- a process that has initialised
    # the GPU must not fork + exec other programs on this pool.
    import multiprocessing
    from multiprocessing import forkserver
    try:
        multiprocessing.get_context("forkserver")
        forkserver.ensure_running()
    except (ValueError, OSError):
        pass


def _run_command(cmd, cwd, env, out_path, timeout):
    import subprocess
    with open(out_path, "w") as f:
        try:
            r = subprocess.run(cmd, cwd=cwd, env=env, stdout=f, stderr=subprocess.STDOUT, timeout=timeout)
            f.write("\n[exit code %d]\n" % r.returncode)
        except subprocess.TimeoutExpired:
            f.write("\n[timed out]\n")


def run_in_clean_child(cmd, out_path, timeout=600, env=None):
    """Runs `cmd` as a child of the fork server started in pytest_configure (not of this process); returns its output."""
    import multiprocessing
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=_run_command, args=(cmd, ROOT, dict(env or os.environ), out_path, timeout))
    p.start()
    p.join(timeout + 30)
    if p.is_alive():
        p.terminate()
    return open(out_path).read() if os.path.exists(out_path) else ""


@pytest.fixture(scope="session")
def fray():
    import fray_amd
    return fray_amd


@pytest.fixture(scope="session")
def abi():
    from fray_amd import abi
    return abi


@pytest.fixture(scope="session")
def oracle(abi):
    """The CPU oracle: test infrastructure, the checker for the HIP path."""
    from oracle.oracle import Oracle
    return Oracle(abi)


def open_scene(fray, name, W=None, H=None, **over):
    s = fray.Scene.parseScene(os.path.join(SCENES, name))
    if W:
        s.settings.frameWidth, s.settings.frameHeight = W, H
    for k, v in over.items():
        if hasattr(s.settings, k):
            setattr(s.settings, k, v)
        elif hasattr(s.camera, k):
            setattr(s.camera, k, v)
        else:
            raise AttributeError(k)
    return s


@pytest.fixture(scope="session")
def gpu(fray):
    rc = fray.lib.frayhip_init(0)
    if rc != 0:
        pytest.fail("frayhip_init(0) failed on a GPU-marked test: %s" % fray.lib.frayhip_last_error())
    return 0
