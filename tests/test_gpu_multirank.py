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
    assert lib.frayhip_gather_buckets(comm, frame.data_ptr(), 200, 130, 3, 1, None) != 0      # root outside the world
    assert lib.frayhip_gather_buckets(comm, None, 200, 130, 3, 0, None) != 0
    lib.frayhip_comm_destroy(comm)
    bad = C.c_void_p()
    assert lib.frayhip_comm_create(None, 0, 2, C.byref(bad)) != 0                              # a world of two needs the id
    assert lib.frayhip_comm_create(bytes(ident), 2, 2, C.byref(bad)) != 0                      # rank outside the world
