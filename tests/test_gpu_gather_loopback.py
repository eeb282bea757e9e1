"""frayhip_gather_buckets with world > 1, EXECUTED (fray_amd/csrc/capi_comm.hip: pack -> ncclSend on the peers; grouped ncclRecv at accumulated
offsets -> per-peer unpack on the root).  RCCL refuses two ranks on one device, and the GPU box has one: the library is told, through
FRAYHIP_RCCL_LIBRARY, to bind tests/native/librccl_loopback.so instead -- the ten entry points over files in /dev/shm -- so the library's own branch
runs unchanged with 2, 3 and 8 ranks.  What the reference gets for free from one shared `vfb` (src/main.cpp:360,404).

  * ranks as threads of one process (tests/gather_loopback_ranks.py): worlds 2 / 3 / 8, root 0 and root != 0, 1920x1080 and 4096x4096 (ragged
    shares), a frame with cut edge buckets, 1 / 2 / 3 channels, one communicator reused across gathers and streams;
  * ranks as processes: `bench.py --gpus N --backend gloo --library-gather --check` (the bench's own N > 1 path, gathered frame == single-rank frame,
    ranks_seen_by_rccl == N) and examples/fray_render_mgpu with N ranks (the BMP must be examples/fray_render's, byte for byte)."""
import json
import os
import socket
import sys

import pytest

from conftest import ROOT, run_in_clean_child

pytestmark = pytest.mark.gpu
LOOPBACK = os.path.join(ROOT, "tests", "native", "librccl_loopback.so")


def loop_env(**more):
    if not os.path.exists(LOOPBACK):
        pytest.fail("tests/native/librccl_loopback.so is not built (make)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FRAYHIP_RCCL_LIBRARY=LOOPBACK, FRAY_LOOPBACK_TIMEOUT_S="240", HSA_ENABLE_IPC_MODE_LEGACY="0", **more)
    return env


@pytest.mark.parametrize("world", [2, 3, 8])
def test_library_gather_executes_with_thread_ranks(tmp_path, world):
    out = run_in_clean_child([sys.executable, os.path.join(ROOT, "tests", "gather_loopback_ranks.py"), str(world)], str(tmp_path / "ranks.log"),
                             timeout=900, env=loop_env())
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-3000:]
    r = json.loads(lines[-1])
    assert r["library"] == LOOPBACK, r["library"]                     # the library bound the stand-in, not the RCCL PyTorch maps
    assert r["ranks_seen"] == [world] * world
    assert not r["errors"], r["errors"]
    assert all(c["equal"] is True for c in r["cases"]), r["cases"]
    assert {c["root"] for c in r["cases"]} != {0} and {c["channels"] for c in r["cases"]} == {1, 2, 3}
    assert r["files_left"] == [] and r["ok"] is True and "[exit code 0]" in out


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world,workload", [(2, "cornell_pt64"), (3, "cornell_pt64"), (3, "dragon_primary")])
def test_bench_ranks_exchange_through_the_library_branch(tmp_path, world, workload):
    """bench.py's N > 1 path as processes, with the exchange inside the library (colour frames: 3 channels; hit-record frames: 1 and 2 channels)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0",
           "--backend", "gloo", "--library-gather", "--check", "--no-cpu-baseline", "--workload", workload]
    out = run_in_clean_child(cmd, str(tmp_path / "bench.log"), timeout=900, env=loop_env(OMP_NUM_THREADS="1"))
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-3000:]
    r = json.loads(lines[-1])
    assert r["n_gpus"] == world and r["gathered_frame_equals_single_rank_frame"] is True, out[-2000:]
    assert r["config"]["gather"].startswith("frayhip_gather_buckets") and LOOPBACK in r["config"]["gather"]
    assert r["config"]["ranks_seen_by_rccl"] == world


@pytest.mark.parametrize("world", [2, 3])
def test_mgpu_example_ranks_write_the_single_gpu_picture(tmp_path, world):
    one, many = os.path.join(ROOT, "examples", "fray_render"), os.path.join(ROOT, "examples", "fray_render_mgpu")
    if not (os.path.exists(one) and os.path.exists(many)):
        pytest.fail("examples are not built (make)")
    scene = os.path.join(ROOT, "scenes", "cornell_box.fray")
    a, b = tmp_path / "one.bmp", tmp_path / "mgpu.bmp"
    l1 = run_in_clean_child([one, scene, str(a), "200", "150", "4"], str(tmp_path / "one.log"), timeout=300)
    assert "[exit code 0]" in l1, l1
    l2 = run_in_clean_child([many, scene, str(b), "200", "150", "4", str(world)], str(tmp_path / "mgpu.log"), timeout=300,
                            env=loop_env(FRAY_RENDER_MGPU_ONE_DEVICE="1"))
    assert "[exit code 0]" in l2 and "on %d GPUs" % world in l2, l2
    assert open(a, "rb").read() == open(b, "rb").read()
