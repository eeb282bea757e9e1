#!/usr/bin/env python3
"""Benchmark of the hot path: one step = one frame of cornell_box.fray at 1920x1080, 64 spp, path
traced (BASELINE.json configs[2], the configuration the metric is quoted on), rendered by the HIP
library with the scene resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

N > 1 is strong scaling of the same frame: the reference's 48x48 buckets are dealt round-robin to
the ranks (no data-path collective while rendering), then one RCCL gather of the packed buckets to
rank 0, inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, W, H, overrides, description)
    "cornell_pt64": ("cornell_box.fray", 1920, 1080, dict(gi=1, numPaths=64),
                     "cornell_box.fray 1920x1080 64spp path trace, maxTraceDepth 6 (BASELINE configs[2])"),
    "smallpt_pt64": ("smallpt.fray", 1920, 1080, dict(gi=1, numPaths=64), "smallpt.fray 1920x1080 64spp path trace"),
    "boxed_whitted": ("boxed.fray", 1920, 1080, dict(wantAA=0), "boxed.fray 1920x1080 1spp Whitted (KD meshes, 32 shadow rays/hit)"),
    "forest_dof256": ("forest.fray", 1920, 1080, dict(wantAA=0, dof=1, numDOFSamples=256, interactive=0),
                      "forest.fray 1920x1080 DOF 256spp Whitted (BASELINE configs[3] on one GPU)"),
    "zaphod_whitted": ("zaphod.fray", 1920, 1080, dict(wantAA=0, dof=0), "zaphod.fray 1920x1080 1spp Whitted (BASELINE configs[1])"),
    "forest_dof16": ("forest.fray", 1920, 1080, dict(wantAA=0, dof=1, numDOFSamples=16, interactive=0), "forest.fray 1920x1080 DOF 16spp Whitted"),
    "smallpt_4k_pt1024": ("smallpt.fray", 4096, 4096, dict(gi=1, numPaths=1024),
                          "smallpt.fray 4096x4096 1024spp path trace, maxTraceDepth 8 (BASELINE configs[4] on one GPU; use --steps 1 --warmup 0)"),
    "dragon_primary": ("hw9/dragon.fray", 1920, 1080, dict(wantAA=0), "hw9/dragon.fray 1920x1080 primary rays (100k-triangle KD)"),
    "forest_primary": ("forest.fray", 1920, 1080, dict(wantAA=0, interactive=0), "forest.fray 1920x1080 primary rays (closest hit only)"),
    "boxed_primary": ("boxed.fray", 1920, 1080, dict(wantAA=0), "boxed.fray 1920x1080 primary rays (closest hit only)"),
    "smallpt_whitted": ("smallpt.fray", 1920, 1080, dict(gi=0, wantAA=0), "smallpt.fray 1920x1080 1spp Whitted (mirror + glass recursion: k_whitted)"),
    "bokeh_dof": ("hw10/bokeh.fray", 640, 480, dict(), "hw10/bokeh.fray 640x480 DOF 45spp Whitted as shipped (Cube - Cube CSG floor, Layered over a mirror, Phong mesh: k_whitted, CSG variant)"),
    "dragon_whitted": ("hw9/dragon.fray", 1920, 1080, dict(wantAA=0), "hw9/dragon.fray 1920x1080 1spp Whitted (glossy floor: k_whitted)"),
    # a scene of this repository (tests/scenes): CsgOp trees three levels deep over a CSG slab floor, a mesh operand, transformed nodes
    "csg_nested_whitted": ("../tests/scenes/csg_nested.fray", 960, 720, dict(wantAA=0), "tests/scenes/csg_nested.fray 960x720 1spp Whitted (nested CsgOps: k_wh_shade / k_wh_visible, CSG variants)"),
    "csg_nested_pt16": ("../tests/scenes/csg_nested.fray", 960, 720, dict(gi=1, numPaths=16), "tests/scenes/csg_nested.fray 960x720 16spp path trace (nested CsgOps: k_pt_bounce / k_pt_shadow, CSG variants)"),
}


def open_scene(fray_amd, name, W, H, over):
    s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", name))
    s.settings.frameWidth, s.settings.frameHeight = W, H
    for k, v in over.items():
        if hasattr(s.settings, k):
            setattr(s.settings, k, v)
        else:
            setattr(s.camera, k, v)
    return s


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by a cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(fray_amd, abi, wl, seed, target_seconds=12.0, gpu_frame=None, other_frames=None):
    """The oracle (a scalar C++ port of the reference path) timed on this box's host cores on a
    bounded sample of the same workload: every `stride`-th 48x48 bucket of the same frame."""
    from oracle.oracle import Oracle
    orc = Oracle(abi)
    name, W, H, over, _ = wl
    s = open_scene(fray_amd, name, W, H, over)
    threads = min(usable_cores(), 64)
    nb = ((W - 1) // 48 + 1) * ((H - 1) // 48 + 1)
    # calibrate on `threads` buckets spread over the frame, then size the sample for ~target_seconds
    mode = abi.MODE_RENDER
    cal_stride = max(1, nb // max(1, min(nb, threads)))
    t0 = time.time()
    orc.render(s.desc, mode, seed=seed, bucket_first=0, bucket_stride=cal_stride, threads=threads)
    dt = max(time.time() - t0, 1e-3)
    per_bucket = dt / len(range(0, nb, cal_stride))
    want = int(max(threads, min(nb, target_seconds / per_bucket)))
    stride = max(1, nb // want)
    t0 = time.time()
    cpu_img, st = orc.render(s.desc, mode, seed=seed, bucket_first=0, bucket_stride=stride, threads=threads)
    dt = time.time() - t0
    parity = None
    other_parity = {}
    if gpu_frame is not None:
        # the buckets the oracle just rendered, against the same pixels of the frame the GPU was timed on
        import numpy as np
        from fray_amd import tiles
        mask = np.zeros((H, W), bool)
        for b in range(0, nb, stride):
            bx, by = tiles.bucket_xy(W, H, b)
            mask[by * 48:by * 48 + 48, bx * 48:bx * 48 + 48] = True
        d = gpu_frame[mask].astype(np.float64) - cpu_img[mask]
        def against_oracle(img):
            d = img[mask].astype(np.float64) - cpu_img[mask]
            return {"pixels": int(mask.sum()), "rms_per_channel": [float(v) for v in np.sqrt((d ** 2).mean(axis=0))],
                    "max_abs": float(np.abs(d).max()), "tolerance": 1e-4,
                    "bit_identical_pixels": float((img[mask] == cpu_img[mask]).all(axis=1).mean())}
        parity = against_oracle(gpu_frame)
        # frames of the same workload rendered in another arithmetic mode (option "fp_contract"): the same buckets against the same oracle pixels
        for k, img in (other_frames or {}).items():
            other_parity[k] = against_oracle(img)
    rays = st["closest_rays"] + st["shadow_rays"]
    n_b = len(range(0, nb, stride))
    # single-thread figure on a smaller sample (~4 s)
    stride1 = max(1, int(nb / max(1.0, 4.0 / (per_bucket * threads))))
    t0 = time.time()
    _, st1 = orc.render(s.desc, mode, seed=seed, bucket_first=0, bucket_stride=stride1, threads=1)
    dt1 = max(time.time() - t0, 1e-6)
    s.close()
    ref_cmp = reference_object_code_rate(fray_amd, abi, orc, wl, seed)
    return {**({"reference_object_code": ref_cmp} if ref_cmp else {}), **({"gpu_frame_vs_oracle_on_the_sample": parity} if parity else {}),
            **({"other_frames_vs_oracle_on_the_sample": other_parity} if other_parity else {}),
            "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "%d of %d buckets (every %d-th 48x48 bucket) of the same frame, all spp, %.1f s, %d threads" % (n_b, nb, stride, dt, threads),
            "frame_ms_extrapolated": dt * 1e3 * nb / n_b,
            "one_thread_mrays_per_s": (st1["closest_rays"] + st1["shadow_rays"]) / dt1 / 1e6,
            "one_thread_sample": "%d buckets, %.1f s" % (len(range(0, nb, stride1)), dt1)}


REF_TIMER = r"""
import ctypes as C, sys, time
import numpy as np
lib = C.CDLL(sys.argv[1])
lib.ref_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
lib.ref_render.argtypes = [C.c_void_p, C.c_uint]
W, H = int(sys.argv[3]), int(sys.argv[4])
assert lib.ref_load(sys.argv[2].encode(), W, H, sys.argv[5].encode()) == 0
img = np.zeros((H, W, 3), np.float32)
t0 = time.time()
lib.ref_render(img.ctypes.data, int(sys.argv[6]))
sys.stderr.write("REFTIME %.6f\n" % (time.time() - t0))
"""


def reference_object_code_rate(fray_amd, abi, orc, wl, seed, W=96, H=96):
    """How the port compares with the reference's own code: oracle/_ref (the reference's geometry / shader /
    light / camera object code under the restated sample loop, oracle/ref_glue.cpp) and the oracle render the
    same small frame of this workload's scene on ONE thread; both produce the same picture, so the oracle's ray
    count applies to both.  Skipped when oracle/_ref was not built (it needs the reference tree at build time)."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_ref", "libfray_ref.so")
    name, _, _, over, _ = wl
    if not os.path.exists(so) or "dof" in over:
        return None
    ov = ";".join("%s=%s" % (k, v) for k, v in over.items())
    try:
        r = subprocess.run([sys.executable, "-c", REF_TIMER, so, os.path.join(ROOT, "scenes", name), str(W), str(H), ov, str(seed)],
                           capture_output=True, text=True, timeout=120, env={**os.environ, "OMP_NUM_THREADS": "1"})
        t_ref = float([l for l in r.stderr.splitlines() if l.startswith("REFTIME ")][-1].split()[1])
    except (subprocess.SubprocessError, IndexError, ValueError, OSError):
        return None
    s = open_scene(fray_amd, name, W, H, over)
    t0 = time.time()
    _, st = orc.render(s.desc, abi.MODE_RENDER, seed=seed, threads=1)
    t_port = max(time.time() - t0, 1e-6)
    s.close()
    rays = st["closest_rays"] + st["shadow_rays"]
    return {"frame": "%dx%d, same scene and spp" % (W, H), "threads": 1,
            "reference_mrays_per_s": rays / max(t_ref, 1e-6) / 1e6, "port_mrays_per_s": rays / t_port / 1e6}


def self_launch(n):
    """`python bench.py --gpus N ...` without a launcher: runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>` as a child process (the command the driver uses), passes its output through and returns its exit
    status.  The port is one the kernel just handed out as free."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # RCCL / tensor sharing across processes on this driver: dmabuf IPC only
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cornell_pt64", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--spp-chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the serialised pass that times every launch alone (path tracing): durations then come from the timed region")
    ap.add_argument("--check", action="store_true", help="N > 1: also render the whole frame on rank 0 and require the gathered frame to equal it")
    ap.add_argument("--torch-gather", action="store_true", help="N > 1: exchange through torch.distributed.gather instead of frayhip_gather_buckets")
    ap.add_argument("--shard-of", type=int, default=0, help="diagnosis on one GPU: render only ONE rank's share of an N-rank run (buckets r mod N, r = --shard-rank), no exchange")
    ap.add_argument("--shard-rank", type=int, default=0, help="with --shard-of N: whose share (tools/shard_balance.py measures every rank's)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the real multi-GPU run) or gloo (rehearsal: N ranks sharing GPU 0)")
    ap.add_argument("--arith", default="exact", choices=["exact", "contracted"],
                    help="exact (default): the reference's arithmetic everywhere, every pixel the oracle's bit for bit.  contracted: option fp_contract = 1 for the timed region "
                    "(path tracing past a sample's first closest hit on kernels built with fused multiply-adds; colour within 1e-4 RMS).  The default run reports the "
                    "contracted figures too, in a `contracted` object beside the exact line")
    ap.add_argument("--no-contracted", action="store_true", help="skip the second timed region that renders the same frame with option fp_contract = 1 (profiling passes: one arithmetic per run)")
    ap.add_argument("--library-gather", action="store_true", help="with --backend gloo: still exchange through frayhip_gather_buckets (the library binds whatever RCCL "
                    "FRAYHIP_RCCL_LIBRARY names -- on a one-GPU box the test suite's loopback stand-in, since RCCL itself refuses two ranks on one device)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` typed like the N = 1 command: start the N ranks ourselves, as CHILD processes of a launcher that is a
        # child of this process (which has not touched the GPU and never will: nothing is exec'ed over it), relay rank 0's JSON line and
        # leave with the launcher's status.
        raise SystemExit(self_launch(args.gpus))

    import torch
    import fray_amd
    from fray_amd import abi

    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d: WORLD_SIZE is 1 although RANK is set; launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if args.backend == "gloo":
        local_rank = 0                      # rehearsal on a one-GPU box: every rank drives GPU 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    rc = fray_amd.lib.frayhip_init(local_rank)
    assert rc == 0, fray_amd.lib.frayhip_last_error()

    wl = WORKLOADS[args.workload]
    name, W, H, over, desc_text = wl
    scene = open_scene(fray_amd, name, W, H, over)
    scene.beginRender()
    mode = abi.MODE_PRIMARY_ID if args.workload.endswith("_primary") else abi.MODE_RENDER
    if args.arith == "contracted":
        if not scene.settings.gi:
            raise SystemExit("bench.py --arith contracted: only the path tracer has contracted kernels")
        scene.set_option("fp_contract", 1)

    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    ids = torch.zeros((H, W), dtype=torch.int32, device=dev) if mode == abi.MODE_PRIMARY_ID else None
    dists = torch.zeros((H, W), dtype=torch.float64, device=dev) if mode == abi.MODE_PRIMARY_ID else None
    lib = fray_amd.lib
    from fray_amd import tiles
    # The one exchange step.  On the real multi-GPU run (--backend nccl) it happens inside the library: frayhip_gather_buckets, grouped RCCL
    # send/recv peer -> root over xGMI.  If that cannot be set up on this host every rank falls back TOGETHER to the same step over
    # torch.distributed, and the JSON line says which transport ran.  --backend gloo (ranks sharing one GPU) stages through the host.
    gatherer = gather_ids = gather_dist = None
    transport = None
    libgather = None
    if world > 1:
        ok = 0
        if (args.backend == "nccl" or args.library_gather) and not args.torch_gather:
            try:
                libgather = tiles.LibraryGather(rank, world, dist)
                ok = 1
            except Exception as e:                 # noqa: BLE001 -- any failure here means "use the other transport"
                sys.stderr.write("rank %d: library gather unavailable (%s)\n" % (rank, e))
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            # one exchange of the (still empty) frame before anything is timed: a transport that raises on its first use is replaced
            # by the other one on EVERY rank, like one that cannot be set up
            try:
                libgather.gather(frame)
                torch.cuda.synchronize()
            except Exception as e:                 # noqa: BLE001
                sys.stderr.write("rank %d: library gather failed on its first exchange (%s)\n" % (rank, e))
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            gatherer = gather_ids = gather_dist = libgather
            transport = "frayhip_gather_buckets (RCCL ncclSend/ncclRecv, peer -> root; bound from %s)" % (lib.frayhip_comm_library().decode() or "?")
        else:
            stage = args.backend != "nccl"
            gatherer = tiles.TileGather(W, H, 3, rank, world, dev, dist, stage_host=stage)
            # hit-record frames gather the same way: int32 ids as one 4-byte channel, f64 distances as two
            gather_ids = tiles.TileGather(W, H, 1, rank, world, dev, dist, stage_host=stage) if ids is not None else None
            gather_dist = tiles.TileGather(W, H, 2, rank, world, dev, dist, stage_host=stage) if ids is not None else None
            transport = "torch.distributed.gather (%s)" % args.backend
    rdev = dev if args.backend == "nccl" else torch.device("cpu")     # where small reduction tensors live

    def stream_ptr():
        return torch.cuda.current_stream().cuda_stream

    def step(stats=False):
        shard = args.shard_of > 1 and world == 1
        st = scene.render_device(frame.data_ptr(), seed=args.seed, bucket_first=args.shard_rank % args.shard_of if shard else rank, bucket_stride=args.shard_of if shard else world,
                                 spp_chunk=args.spp_chunk, stats=stats, stream=stream_ptr(), mode=mode,
                                 d_id_ptr=ids.data_ptr() if ids is not None else None,
                                 d_dist_ptr=dists.data_ptr() if dists is not None else None)
        if world > 1 and mode == abi.MODE_RENDER:
            # the one exchange step: packed buckets -> rank 0 (peer-to-root sends over xGMI), then untile
            gatherer.gather(frame)
        elif world > 1:
            gather_ids.gather(ids.view(torch.float32).view(H, W, 1))          # bit patterns travel unchanged
            gather_dist.gather(dists.view(torch.float32).view(H, W, 2))
        return st

    # counters + algorithmic bytes of one frame (instrumented kernels, untimed)
    st_counts = step(stats=True)
    counts = torch.tensor([st_counts["closest_rays"], st_counts["shadow_rays"], st_counts["samples"], st_counts["alg_bytes_trace"]],
                          dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(counts)
    rays_total = float(counts[0] + counts[1])

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trace_ms = 0.0
    trace_launches = 0
    shadow_ms = 0.0
    shadow_launches = 0
    kernels_ms = 0.0
    for _ in range(args.steps):
        st = step()
        trace_ms += st["ms_trace"]
        trace_launches += st["trace_launches"]
        shadow_ms += st["ms_shadow"]
        shadow_launches += st["shadow_launches"]
        kernels_ms += st["ms_kernels"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    ms_per_step = elapsed * 1e3 / args.steps

    check = None
    if args.check and world > 1 and mode == abi.MODE_PRIMARY_ID:
        step()
        if rank == 0:
            wi, wd = torch.zeros_like(ids), torch.zeros_like(dists)
            scene.render_device(frame.data_ptr(), seed=args.seed, bucket_first=0, bucket_stride=1, stream=stream_ptr(), mode=mode,
                                d_id_ptr=wi.data_ptr(), d_dist_ptr=wd.data_ptr())
            torch.cuda.synchronize()
            check = bool(torch.equal(wi, ids) and torch.equal(wd, dists))
        dist.barrier()
    if args.check and world > 1 and mode == abi.MODE_RENDER:
        step()
        if rank == 0:
            whole = torch.zeros_like(frame)
            scene.render_device(whole.data_ptr(), seed=args.seed, bucket_first=0, bucket_stride=1, spp_chunk=args.spp_chunk,
                                stream=stream_ptr(), mode=mode)
            torch.cuda.synchronize()
            check = bool(torch.equal(whole, frame))
        dist.barrier()

    # Per-kernel launch durations for the roofline.  In the timed region up to four batches of a path-traced frame
    # overlap on four streams, so a launch's event-to-event duration there includes the time it shared the chip
    # (the durations sum to more than the frame).  The roofline therefore takes its durations from a second, SERIALISED
    # pass of the same frames (one batch lane: every launch alone on the chip, HIP events around each launch on the
    # stream it runs on) -- the same mode the kept rocprofv3 kernel trace under profiles/ is collected in.
    serial = None
    serial_note = None
    if rank == 0 and world == 1 and scene.settings.gi and not args.no_serial_pass:
        lanes_before = int(os.environ.get("FRAYHIP_PT_LANES", "4") or 4)       # the library's own default (render_state.hpp) unless the environment presets it
        # A frame that takes seconds (4096 x 4096 x 1024 spp: 13 s) is not rendered twice more: its serialised pass renders the SAME frame with fewer
        # samples per pixel -- the batches, hence the launches, keep their size (a batch's samples per pixel follow from the queue budget and the frame's
        # pixels, not from spp), there are just fewer of them.  Launch counts are then scaled back to the full frame.
        spp_full = scene.samples_per_pixel()
        spp_ser = spp_full
        if ms_per_step > 1500.0 and spp_full > 32:
            spp_ser = 32
        ser_scene = scene
        if spp_ser != spp_full:
            ser_scene = open_scene(fray_amd, name, W, H, dict(over, numPaths=spp_ser))
            ser_scene.beginRender()
            serial_note = "a frame of %d spp instead of %d (same batches, fewer of them); launch counts scaled by %g" % (spp_ser, spp_full, spp_full / spp_ser)

        def ser_step():
            return ser_scene.render_device(frame.data_ptr(), seed=args.seed, bucket_first=0, bucket_stride=1, spp_chunk=args.spp_chunk, stream=stream_ptr(), mode=mode)
        ser_scene.set_option("pt_lanes", 1)
        ser_step()
        torch.cuda.synchronize()
        n_ser = max(1, min(args.steps, 3))
        t1 = time.perf_counter()
        acc = {"ms_trace": 0.0, "trace_launches": 0, "ms_shadow": 0.0, "shadow_launches": 0, "ms_kernels": 0.0}
        for _ in range(n_ser):
            st = ser_step()
            for k in acc:
                acc[k] += st[k]
        torch.cuda.synchronize()
        scale = spp_full / spp_ser
        serial = {k: v / n_ser * scale for k, v in acc.items()}
        serial["ms_per_step"] = (time.perf_counter() - t1) * 1e3 / n_ser * scale
        ser_scene.set_option("pt_lanes", lanes_before)
        if ser_scene is not scene:
            ser_scene.close()

    # The same frame with option "fp_contract" = 1 (north_star bounds shaded colour by 1e-4 RMS, only PRIMARY hit records by bits): timed like the region above,
    # counted by its own instrumented pass, and kept for the parity figures of cpu_baseline below.  The default line stays the exact one.
    contracted = None
    exact_frame = frame
    if rank == 0 and world == 1 and scene.settings.gi and mode == abi.MODE_RENDER and args.arith == "exact" and args.shard_of <= 1 and not args.no_contracted:
        exact_frame = frame.clone()
        scene.set_option("fp_contract", 1)
        st_c = step(stats=True)
        for _ in range(min(args.warmup, 2)):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms_c = (time.perf_counter() - t1) * 1e3 / args.steps
        scene.set_option("fp_contract", 0)
        rays_c = float(st_c["closest_rays"] + st_c["shadow_rays"])
        contracted = {"option": "fp_contract = 1: k_pt_bounce after a sample's first closest hit and every k_pt_shadow launch run the kernels of render_contract.hip (-ffp-contract=fast, "
                                "reciprocal / rsqrt with two refinement steps, plain-double sin / cos); primary hits and first bounces stay exact",
                      "contracted_launches_per_frame": scene.get_option("contracted_launches"),
                      "ms_per_step": ms_c, "value": rays_c / (ms_c * 1e-3) / 1e6, "unit": "Mrays/s", "rays_per_frame": rays_c, "steps": args.steps,
                      "speedup_vs_exact": ms_per_step / ms_c, "frame": frame.clone()}

    if rank == 0:
        from tools.source_hash import source_hash
        src = source_hash()
        # the kernel whose launches ms_trace / alg_flops_trace describe, and the one behind ms_shadow / alg_flops_shadow
        wavefront_whitted = mode == abi.MODE_RENDER and not scene.settings.gi and scene.get_option("whitted_path") in (1, 2)
        kern = {abi.MODE_PRIMARY_ID: "k_primary"}.get(mode, "k_pt_bounce" if scene.settings.gi else ("k_wh_shade" if wavefront_whitted else "k_whitted"))
        kern_shadow = "k_pt_shadow" if scene.settings.gi else "k_wh_visible"
        # launch durations: serialised pass for path tracing, the timed region itself otherwise (one kernel, one stream)
        tr_ms = serial["ms_trace"] if serial else trace_ms / args.steps
        tr_n = serial["trace_launches"] if serial else trace_launches / args.steps
        avg_launch_ms = tr_ms / max(1, tr_n)
        flops_per_launch = st_counts["alg_flops_trace"] / max(1, tr_n)
        bytes_per_launch = st_counts["alg_bytes_trace"] / max(1, tr_n)
        tf = flops_per_launch / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        FP64_PEAK = 39.3        # TFLOP/s: MI355X vector FP64 78.6 TFLOP/s counts an FMA as two; the reference arithmetic has no FMA (one operation per lane per issue)
        out = {
            "source_hash": src,          # of the device code that ran (tools/source_hash.py): kept measurement sets must agree on it
            "metric": "Mrays/s (closest-hit + shadow rays) at %dx%d, %dspp %s" % (
                W, H, scene.samples_per_pixel(),
                "primary rays" if mode == abi.MODE_PRIMARY_ID else ("path trace" if scene.settings.gi else "Whitted")),
            "value": rays_total / (ms_per_step * 1e-3) / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "reference scene file scenes/%s (unchanged), RNG contract seed %d" % (name, args.seed),
            "config": {"workload": desc_text, "width": W, "height": H, "spp": scene.samples_per_pixel(),
                       "rays_per_frame": rays_total, "camera_samples_per_frame": float(counts[2]),
                       "arith": args.arith,
                       **({"whitted_path": {0: "k_whitted (recursive shaders)", 1: "k_wh_shade -> k_wh_visible -> k_wh_gather", 2: "k_wh_shade, fused (visible() in place)"}[scene.get_option("whitted_path")]}
                          if mode == abi.MODE_RENDER and not scene.settings.gi else {}),
                       "parallelism": "tiles%d" % world if world > 1 else ("rank %d's share of tiles%d, no exchange (diagnosis)" % (args.shard_rank, args.shard_of) if args.shard_of > 1 else "single-gpu"),
                       "frame_ms": ms_per_step, "msamples_per_s": float(counts[2]) / (ms_per_step * 1e-3) / 1e6,
                       # glossy fans drawn and traced ahead in the last timed frame (frayhip_scene_get_option; zero unless the scene has such fans and no sampling light)
                       "speculative_fans": {k: scene.get_option(k) for k in ("fans_filed", "fan_children", "fan_children_looked_up", "fans_given_up")}},
            # What bounds the dominant kernel is FP64 vector issue, not HBM: algorithmic FP64
            # operations per launch (SURVEY 8d operation counts x this frame's work counters) / average launch duration.
            "roofline": {"why_not_hbm": "the contract's HBM roofline does not bound this path: SURVEY 8(d)'s algorithmic bytes are node transforms and triangle records that the scalar cache serves "
                                        "(roofline_hbm.frac comes out ABOVE 1), measured HBM traffic is `traffic` (about 0.2 of peak); what bounds the kernels is FP64 vector issue",
                         "bound": "fp64_valu", "kernel": kern, "achieved": tf, "peak": FP64_PEAK, "unit": "TFLOP/s", "frac": tf / FP64_PEAK,
                         "alg_flops_per_launch": flops_per_launch, "avg_launch_ms": avg_launch_ms, "launches_per_step": tr_n,
                         "sum_launch_ms_per_step": tr_ms,
                         "durations_from": ("serialised pass (pt_lanes = 1), %.2f ms per frame%s" % (serial["ms_per_step"], (": " + serial_note) if serial_note else "")) if serial else "the timed region",
                         "traffic": None, "counters": None, "source_hash": src},
            # The contract's HBM figure: SURVEY 8(d) algorithmic bytes over the same durations.  Those bytes are node transforms and
            # triangle records served by the scalar cache / L2 (the scene is a few KB), so this is NOT a DRAM rate and may exceed the peak.
            "roofline_hbm": {"bound": "hbm", "kernel": kern, "achieved": bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0,
                             "peak": 8000.0, "unit": "GB/s", "alg_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_launch_ms,
                             "note": "algorithmic bytes per SURVEY 8(d), cache-served; measured HBM traffic is roofline.traffic"},
            "kernel_ms_per_step": kernels_ms / args.steps,
            "launch_ms_sums_per_step": {"trace": trace_ms / args.steps, "shadow": shadow_ms / args.steps,
                                        "launches": [trace_launches / args.steps, shadow_launches / args.steps],
                                        "note": "timed region; launches of up to four batch lanes overlap, so these sums exceed ms_per_step"},
        }
        out["roofline_hbm"]["frac"] = out["roofline_hbm"]["achieved"] / 8000.0
        # The same ceiling for the whole frame as it ships (batch lanes overlapping, seeding and resolves included): every kernel's algorithmic FP64 operations
        # over the frame time.  The kernel-level figure above is per launch ALONE on the chip; four lanes fill each other's gaps, so the frame sits higher.
        frame_flops = st_counts["alg_flops_trace"] + st_counts["alg_flops_shadow"]
        out["roofline"]["frame_level"] = {"alg_flops_per_frame": frame_flops, "achieved": frame_flops / (ms_per_step * 1e-3) / 1e12, "unit": "TFLOP/s",
                                          "frac": frame_flops / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK,
                                          "note": "all kernels' algorithmic FP64 operations / ms_per_step (timed region, batch lanes overlapping)"}
        sh_ms = serial["ms_shadow"] if serial else shadow_ms / args.steps
        sh_n = serial["shadow_launches"] if serial else shadow_launches / args.steps
        if sh_n and sh_ms > 0:
            sms = sh_ms / sh_n
            sf = st_counts["alg_flops_shadow"] / sh_n
            out["roofline_shadow_kernel"] = {"bound": "fp64_valu", "kernel": kern_shadow, "achieved": sf / (sms * 1e-3) / 1e12, "peak": FP64_PEAK, "unit": "TFLOP/s",
                                             "frac": sf / (sms * 1e-3) / 1e12 / FP64_PEAK, "alg_flops_per_launch": sf, "avg_launch_ms": sms,
                                             "launches_per_step": sh_n, "sum_launch_ms_per_step": sh_ms,
                                             "alg_bytes_per_launch": st_counts["alg_bytes_shadow"] / sh_n}
            if sh_ms > tr_ms:         # the shadow-ray kernel is the dominant one of this workload (e.g. boxed: 32 of 33 rays): it is THE roofline
                out["roofline"], out["roofline_shade_kernel"] = dict(out["roofline_shadow_kernel"], traffic=None, counters=None, source_hash=src,
                                                                     durations_from=out["roofline"]["durations_from"]), out["roofline"]
                del out["roofline_shadow_kernel"]
        if check is not None:
            out["gathered_frame_equals_single_rank_frame"] = check
        if transport:
            out["config"]["gather"] = transport
            # how many ranks RCCL itself sees in the communicator the frame was gathered on (null: the exchange did not run on RCCL)
            out["config"]["ranks_seen_by_rccl"] = gatherer.ranks_seen_by_rccl() if gatherer is libgather and libgather is not None else None
        # Counter-derived figures (HBM traffic, VALU issue and lane utilisation) cannot be collected by this process: they come
        # from rocprofv3 --pmc passes of this same command (tools/profile_workload.sh -> profiles/pmc_latest_<workload>.json) and are
        # printed only when that profile was taken on the same device code (source hash) and workload; otherwise they stay null.
        pmc_key = args.workload + ("_contracted" if args.arith == "contracted" else "")       # tools/profile_workload.sh profiles one arithmetic per run
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest_%s.json" % pmc_key)
        if not os.path.exists(pmc_path) and args.arith == "exact":
            pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if world == 1 and os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("source_hash") == src and pmc.get("workload") == pmc_key:
                    for key in ("roofline", "roofline_shadow_kernel", "roofline_shade_kernel"):
                        k = pmc["kernels"].get(out[key]["kernel"]) if key in out else None
                        # per-launch figures only describe launches of the profiled size (another queue budget cuts the frame into other launches)
                        if k and k.get("avg_launch_ms") and not 0.75 <= out[key]["avg_launch_ms"] / k["avg_launch_ms"] <= 1.33:
                            out[key]["counters_note"] = "%s was taken with launches of %.3f ms: not this configuration's" % (os.path.relpath(pmc_path, ROOT), k["avg_launch_ms"])
                            k = None
                        if k:
                            out[key]["traffic"] = k.get("hbm_bytes_per_launch")
                            out[key]["traffic_note"] = k.get("hbm_note")
                            out[key]["counters"] = k.get("derived")
                            out[key]["profile_avg_launch_ms"] = k.get("avg_launch_ms")
                            mix = k.get("instruction_mix_per_launch")
                            if mix:
                                # MEASURED beside the estimate: FP64 wave-instructions the kernel issued per launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64), as lane
                                # operations at the measured lane utilisation, next to alg_flops_per_launch (SURVEY 8d operation counts x work counters)
                                lanes = (k.get("derived") or {}).get("lane_utilisation") or 1.0
                                out[key]["fp64_wave_instructions_measured"] = mix["fp64_wave_instructions"]
                                out[key]["fp64_ops_measured"] = mix["fp64_wave_instructions"] * 64.0 * lanes
                                out[key]["instruction_mix"] = {a: b for a, b in mix.items() if a != "fp64_wave_instructions"}
                else:
                    out["roofline"]["counters_note"] = "%s is for source %s / %s: not this build" % (os.path.relpath(pmc_path, ROOT), pmc.get("source_hash"), pmc.get("workload"))
            except (KeyError, ValueError, OSError):
                pass
        cframe = contracted.pop("frame") if contracted else None
        if contracted:
            # vector instructions per 64-ray iteration of the bounce kernel in both arithmetics, from the two kept counter profiles (same source hash only)
            try:
                pe = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest_%s.json" % args.workload)))
                pc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest_%s_contracted.json" % args.workload)))
                if pe.get("source_hash") == src and pc.get("source_hash") == src:
                    for kname in ("k_pt_bounce", "k_pt_shadow"):
                        a, b = pe["kernels"][kname]["per_launch"], pc["kernels"][kname]["per_launch"]
                        contracted.setdefault("valu_wave_instructions_per_launch", {})[kname] = {"exact": a["SQ_INSTS_VALU"], "contracted": b["SQ_INSTS_VALU"],
                                                                                                "ratio": b["SQ_INSTS_VALU"] / a["SQ_INSTS_VALU"]}
                        contracted.setdefault("avg_launch_ms", {})[kname] = {"exact": pe["kernels"][kname].get("avg_launch_ms"), "contracted": pc["kernels"][kname].get("avg_launch_ms")}
            except (KeyError, ValueError, OSError, ZeroDivisionError):
                pass
            d = (cframe.double() - exact_frame.double())
            contracted["vs_exact_frame"] = {"rms_per_channel": [float(v) for v in (d ** 2).mean(dim=(0, 1)).sqrt().cpu()], "max_abs": float(d.abs().max()),
                                            "bit_identical_pixels": float((cframe == exact_frame).all(dim=2).float().mean())}
            out["contracted"] = contracted
            out["value_contracted"] = contracted["value"]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(fray_amd, abi, wl, args.seed, gpu_frame=exact_frame.cpu().numpy(),
                                               other_frames={"contracted": cframe.cpu().numpy()} if cframe is not None else None) if mode == abi.MODE_RENDER else None
            if contracted and out["cpu_baseline"] and "other_frames_vs_oracle_on_the_sample" in out["cpu_baseline"]:
                contracted["vs_oracle_on_the_sample"] = out["cpu_baseline"].pop("other_frames_vs_oracle_on_the_sample")["contracted"]
                contracted["rms_per_channel"] = contracted["vs_oracle_on_the_sample"]["rms_per_channel"]
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        # every rank's work is done and seen by every other rank before a communicator goes away
        torch.cuda.synchronize()
        dist.barrier()
        if libgather is not None:
            libgather.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
