"""The N > 1 path on CPU: world_size 2 (and 3) over gloo.  Each rank renders its own buckets -- here
with the oracle standing in for the GPU, since this checks the partition + pack + gather + unpack
contract, not the kernels -- and rank 0 must end up with exactly the single-rank frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, SCENES, bucket_xy


def np_pack(frame, W, H, rank, world, out):
    f = frame.numpy()
    C = f.shape[-1]
    BW = (W - 1) // 48 + 1
    nb_total = BW * ((H - 1) // 48 + 1)
    o = out.numpy().reshape(-1, 48, 48, C)
    for k, b in enumerate(range(rank, nb_total, world)):
        bx, by = bucket_xy(W, b)
        tile = np.zeros((48, 48, C), np.float32)
        sub = f[by * 48:by * 48 + 48, bx * 48:bx * 48 + 48]
        tile[:sub.shape[0], :sub.shape[1]] = sub
        o[k] = tile


def np_unpack(packed, frame, W, H, rank, world):
    f = frame.numpy()
    C = f.shape[-1]
    BW = (W - 1) // 48 + 1
    nb_total = BW * ((H - 1) // 48 + 1)
    p = packed.numpy().reshape(-1, 48, 48, C)
    for k, b in enumerate(range(rank, nb_total, world)):
        bx, by = bucket_xy(W, b)
        sub = f[by * 48:by * 48 + 48, bx * 48:bx * 48 + 48]
        sub[...] = p[k][:sub.shape[0], :sub.shape[1]]


def worker(rank, world, port, W, H, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import fray_amd
    from fray_amd import abi, tiles
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = fray_amd.Scene.parseScene(os.path.join(SCENES, "cornell_box.fray"))
    s.settings.frameWidth, s.settings.frameHeight, s.settings.numPaths = W, H, 2
    orc = Oracle(abi)
    mine, st = orc.render(s.desc, abi.MODE_RENDER, seed=42, bucket_first=rank, bucket_stride=world, threads=2)
    frame = torch.from_numpy(mine.copy())
    tg = tiles.TileGather(W, H, 3, rank, world, "cpu", dist, pack=np_pack, unpack=np_unpack)
    tg.gather(frame)
    rays = torch.tensor([float(st["closest_rays"] + st["shadow_rays"])], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        full, fst = orc.render(s.desc, abi.MODE_RENDER, seed=42, threads=2)
        np.save(os.path.join(outdir, "ok.npy"), np.array([np.array_equal(frame.numpy(), full),
                                                          float(rays[0]) == fst["closest_rays"] + fst["shadow_rays"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 200, 130), (3, 97, 101)])
def test_bucket_shards_gather_to_the_single_rank_frame(tmp_path, world, W, H):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(worker, args=(world, port, W, H, str(tmp_path)), nprocs=world, join=True)
    ok = np.load(os.path.join(str(tmp_path), "ok.npy"))
    assert ok[0] == 1 and ok[1] == 1


def test_bench_gpus_n_without_a_launcher_starts_the_ranks_itself(tmp_path):
    """`python bench.py --gpus 2 --backend gloo` typed like the N = 1 command (no torch.distributed.run in front): bench.py must start the two ranks
    itself -- as child processes, before anything in it touches a GPU -- and leave with their status.  Here, without a GPU, both ranks get as far as
    asking for their device and fail THERE (not on the launch convention); tests/test_gpu_multirank.py runs the same command to the JSON line."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("the no-GPU half of this check; on a GPU box tests/test_gpu_multirank.py runs the same command to its JSON line from a clean child process")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    err = r.stderr
    assert "starting 2 ranks" in err and "-m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port" in err, err[-2000:]
    assert "must be launched with" not in err
    assert r.returncode != 0                                      # the ranks' failure is the program's exit status
    assert "rank" in err.lower() or "cuda" in err.lower() or "hip" in err.lower(), err[-2000:]
