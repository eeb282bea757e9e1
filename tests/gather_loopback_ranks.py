"""TEST HELPER (run as a program by tests/test_gpu_gather_loopback.py, in a clean child with FRAYHIP_RCCL_LIBRARY naming
tests/native/librccl_loopback.so): N ranks as N THREADS of this process, all on GPU 0, each with its own communicator, frame
and stream, call frayhip_gather_buckets -- the library's real world > 1 branch (pack -> ncclSend; grouped ncclRecv at
accumulated offsets -> per-peer unpack, fray_amd/csrc/capi_comm.hip) -- and the root's frame must equal the frame every
rank's buckets were cut from.  Threads, not processes, so that eight ranks stay one GPU process.

Prints one JSON line: {"world": N, "library": path, "cases": [...], "ok": true}."""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fray_amd  # noqa: E402
from conftest import bucket_xy  # noqa: E402  (the bucket rule restated from include/frayhip.h, not the library's own)


def owner_map(W, H, world):
    BW, BH = (W - 1) // 48 + 1, (H - 1) // 48 + 1
    own = np.zeros((H, W), np.int32)
    for b in range(BW * BH):
        bx, by = bucket_xy(W, b)
        own[by * 48:by * 48 + 48, bx * 48:bx * 48 + 48] = b % world
    return own


def main():
    world = int(sys.argv[1])
    lib = fray_amd.lib
    assert lib.frayhip_init(0) == 0, lib.frayhip_last_error()
    torch.cuda.set_device(0)
    ident = (C.c_char * 128)()
    assert lib.frayhip_comm_unique_id(ident) == 0, lib.frayhip_last_error()
    ident = bytes(ident)
    # (W, H, channels, root): both bench sizes (4096 x 4096: 7 396 buckets, ragged shares for 3 and 8 ranks), a frame whose edge buckets are cut,
    # every channel count the bench uses (1: hit ids, 2: f64 distances, 3: colour), root 0 and root != 0; all on ONE communicator per rank, so the
    # staging buffer grows, is reused and is handed from stream to stream
    cases = [(1920, 1080, 3, 0), (1920, 1080, 1, world - 1), (4096, 4096, 2, 1 % world), (4096, 4096, 3, 0), (1000, 700, 3, world // 2), (50, 40, 3, 0)]
    gen = torch.Generator(device="cuda").manual_seed(1234)
    fulls = [torch.rand((H, W, ch), device="cuda", generator=gen) for (W, H, ch, _) in cases]
    owners = [torch.from_numpy(owner_map(W, H, world)).cuda() for (W, H, _, _) in cases]
    torch.cuda.synchronize()
    results = [None] * len(cases)
    errors = []
    seen = [0] * world
    barrier = threading.Barrier(world)

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            assert lib.frayhip_init(0) == 0
            comm = C.c_void_p()
            rc = lib.frayhip_comm_create(ident, r, world, C.byref(comm))
            assert rc == 0, lib.frayhip_last_error()
            seen[r] = lib.frayhip_comm_ranks(comm)
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            for k, (W, H, ch, root) in enumerate(cases):
                st = streams[k & 1]
                with torch.cuda.stream(st):
                    mine = (owners[k] == r).unsqueeze(-1)
                    frame = torch.where(mine, fulls[k], torch.full_like(fulls[k], float("nan")))       # only this rank's buckets hold the picture
                    rc = lib.frayhip_gather_buckets(comm, frame.data_ptr(), W, H, ch, root, st.cuda_stream)
                    assert rc == 0, lib.frayhip_last_error()
                    st.synchronize()
                    if r == root:
                        results[k] = bool(torch.equal(frame, fulls[k]))
                    else:
                        # a peer's frame is untouched: still its own buckets and nothing else
                        ok = bool(torch.equal(torch.nan_to_num(frame, nan=-1.0), torch.where(mine, fulls[k], torch.full_like(fulls[k], -1.0))))
                        if not ok:
                            errors.append("rank %d: frame changed by a gather it was not the root of (case %d)" % (r, k))
            barrier.wait(timeout=300)
            lib.frayhip_comm_destroy(comm)
        except Exception as e:  # noqa: BLE001
            errors.append("rank %d: %r" % (r, e))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    left = [f for f in os.listdir("/dev/shm") if f.startswith("frayloop_")]
    out = {"world": world, "library": lib.frayhip_comm_library().decode(), "ranks_seen": seen,
           "cases": [{"W": W, "H": H, "channels": ch, "root": root, "equal": results[k]} for k, (W, H, ch, root) in enumerate(cases)],
           "errors": errors, "files_left": left}
    out["ok"] = not errors and all(results) and all(s == world for s in seen)
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    raise SystemExit(main())
