"""The N > 1 path of bench.py end to end on the GPU box: fresh child processes (started from a launcher that never
touched the GPU), `torch.distributed.run --nproc-per-node 2`, both ranks rendering their own buckets with the HIP
path on the one GPU, the exchange step, and --check: rank 0's gathered frame must EQUAL a single-rank render.

The one-GPU box cannot host two RCCL ranks (one device per rank), so the transport here is gloo with host staging;
the RCCL transport inside the library (frayhip_gather_buckets) is covered as far as one rank can take it."""
import ctypes as C
import json
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, run_in_clean_child

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("workload", ["cornell_pt64", "dragon_primary"])
def test_bench_two_ranks_gather_equals_single_rank_frame(tmp_path, workload):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--backend", "gloo", "--check", "--no-cpu-baseline", "--workload", workload]
    out = run_in_clean_child(cmd, str(tmp_path / "bench.log"), timeout=600)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-3000:]
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 2 and r["gathered_frame_equals_single_rank_frame"] is True, out[-2000:]
    assert r["config"]["gather"].startswith("torch.distributed.gather")
    assert r["value"] > 0 and r["scaling"] == "strong"


def test_bench_gpus_2_typed_like_the_single_gpu_command_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` WITHOUT a launcher (the command the driver uses for N = 1, with another N): bench.py starts the ranks itself, as
    children of a launcher that is its own child, and relays rank 0's ONE JSON line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--check", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = run_in_clean_child(cmd, str(tmp_path / "bench.log"), timeout=600, env=env)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["gathered_frame_equals_single_rank_frame"] is True and "[exit code 0]" in out, out[-2000:]
    assert "starting 2 ranks" in out and "torch.distributed.run" in out
    assert r["config"]["ranks_seen_by_rccl"] is None            # a gloo rehearsal on one GPU: the exchange did not run on RCCL, and the line says so


def test_comm_from_nccl_wraps_a_communicator_the_host_owns(fray, gpu):
    """frayhip_comm_from_nccl: a world-of-one ncclComm_t made by the host itself (RCCL through ctypes: the copy PyTorch already mapped) is accepted, reports
    ONE rank seen by RCCL, gathers in place; a world / rank that is not the communicator's own is refused; destroying the wrapper leaves the host's
    communicator alive."""
    import torch
    lib = fray.lib
    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = C.CDLL(name, mode=os.RTLD_NOLOAD | os.RTLD_NOW | os.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        rccl = C.CDLL("librccl.so.1")
    torch.zeros(1, device="cuda")                                # the device is current and initialised

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    nc = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(nc), 1, uid, 0) == 0 and nc.value
    comm = C.c_void_p()
    assert lib.frayhip_comm_from_nccl(nc, 0, 1, C.byref(comm)) == 0, lib.frayhip_last_error()
    assert lib.frayhip_comm_ranks(comm) == 1
    frame = torch.rand((100, 150, 3), device="cuda")
    before = frame.clone()
    assert lib.frayhip_gather_buckets(comm, frame.data_ptr(), 150, 100, 3, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(frame, before)
    lib.frayhip_comm_destroy(comm)
    bad = C.c_void_p()
    assert lib.frayhip_comm_from_nccl(nc, 0, 2, C.byref(bad)) != 0 and b"communicator says" in lib.frayhip_last_error()
    assert lib.frayhip_comm_from_nccl(None, 0, 1, C.byref(bad)) != 0
    # the host's communicator survived the wrapper
    n = C.c_int(0)
    rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    assert rccl.ncclCommCount(nc, C.byref(n)) == 0 and n.value == 1
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclCommDestroy(nc) == 0


def test_library_gather_single_rank_and_argument_checks(fray, gpu):
    """frayhip_comm_* / frayhip_gather_buckets as far as one rank goes: the id comes from RCCL itself, a world of one
    gathers in place, bad arguments are refused."""
    import torch
    lib = fray.lib
    ident = (C.c_char * 128)()
    assert lib.frayhip_comm_unique_id(ident) == 0 and any(bytes(ident))
    comm = C.c_void_p()
    assert lib.frayhip_comm_create(None, 0, 1, C.byref(comm)) == 0
    frame = torch.rand((130, 200, 3), device="cuda")
    before = frame.clone()
    assert lib.frayhip_gather_buckets(comm, frame.data_ptr(), 200, 130, 3, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(frame, before)
    assert lib.frayhip_comm_ranks(comm) == 1
    assert lib.frayhip_gather_buckets(comm, frame.data_ptr(), 200, 130, 3, 1, None) != 0      # root outside the world
    assert lib.frayhip_gather_buckets(comm, None, 200, 130, 3, 0, None) != 0
    lib.frayhip_comm_destroy(comm)
    bad = C.c_void_p()
    assert lib.frayhip_comm_create(None, 0, 2, C.byref(bad)) != 0                              # a world of two needs the id
    assert lib.frayhip_comm_create(bytes(ident), 2, 2, C.byref(bad)) != 0                      # rank outside the world


def test_comm_available_reports_rccl(fray, gpu):
    """frayhip_comm_available: what every rank asks before any rank enters ncclCommInitRank (fray_amd/tiles.py LibraryGather)."""
    assert fray.lib.frayhip_comm_available() == 1


def test_mgpu_example_with_one_rank_writes_the_single_gpu_picture(tmp_path):
    """examples/fray_render_mgpu (one process per GPU, frayhip_gather_buckets as the exchange) with N = 1 must write, byte for byte,
    the BMP examples/fray_render writes: the fork / communicator / gather plumbing around the same frame."""
    one, many = os.path.join(ROOT, "examples", "fray_render"), os.path.join(ROOT, "examples", "fray_render_mgpu")
    if not (os.path.exists(one) and os.path.exists(many)):
        pytest.fail("examples are not built (make)")
    scene = os.path.join(ROOT, "scenes", "cornell_box.fray")
    a, b = tmp_path / "one.bmp", tmp_path / "mgpu.bmp"
    # both programs are children of the fork server (conftest.py), not of this process, which other tests have left GPU-initialised
    l1 = run_in_clean_child([one, scene, str(a), "96", "64", "4"], str(tmp_path / "one.log"), timeout=300)
    assert "[exit code 0]" in l1, l1
    l2 = run_in_clean_child([many, scene, str(b), "96", "64", "4", "1"], str(tmp_path / "mgpu.log"), timeout=300)
    assert "[exit code 0]" in l2, l2
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("workload,spp", [("cornell", 64), ("forest_dof", 16)])
def test_union_of_eight_shards_equals_the_full_frame_at_full_size(fray, gpu, workload, spp):
    """What an 8-rank run computes, on one GPU: the eight bucket shards (b = r mod 8) rendered one after the other into one frame must
    equal, bit for bit, the frame rendered in one call -- 1920x1080, one path-traced and one Whitted (DOF) configuration."""
    import fray_amd
    W, H = 1920, 1080
    if workload == "cornell":
        s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", "cornell_box.fray"))
        s.settings.gi, s.settings.numPaths = 1, spp
    else:
        s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", "forest.fray"))
        s.settings.wantAA, s.settings.interactive = 0, 0
        s.camera.dof, s.camera.numDOFSamples = 1, spp
    s.settings.frameWidth, s.settings.frameHeight = W, H
    s.beginRender()
    whole, _ = s.render(seed=42)
    parts = np.zeros_like(whole)
    for r in range(8):
        s.render(seed=42, bucket_first=r, bucket_stride=8, out=parts)
    assert np.array_equal(whole, parts)
    s.close()


def test_bench_line_contract_single_gpu(tmp_path):
    """`python bench.py` (N = 1, a short run) prints ONE JSON line with the fields the driver reads, the roofline of the dominant
    kernel with non-overlapped launch durations, and the CPU baseline with its in-run parity check."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"]
    out = run_in_clean_child(cmd, str(tmp_path / "bench.log"), timeout=600)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["unit"] == "Mrays/s" and r["dtype"] == "f64" and r["vs_baseline"] is None
    assert "cornell_box.fray 1920x1080 64spp" in r["config"]["workload"]
    assert abs(r["value"] - r["config"]["rays_per_frame"] / (r["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * r["value"]
    rf = r["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step"):
        assert k in rf, k
    assert rf["kernel"] == "k_pt_bounce" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # launch durations come from the serialised pass: they add up to less than that pass's frame time
    assert rf["launches_per_step"] * rf["avg_launch_ms"] <= float(rf["durations_from"].split(",")[1].split()[0])
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    par = cb["gpu_frame_vs_oracle_on_the_sample"]
    assert max(par["rms_per_channel"]) <= 1e-4 and par["bit_identical_pixels"] >= 0.99
