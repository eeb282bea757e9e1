"""Multi-GPU tile exchange: every rank renders the 48x48 buckets b = rank (mod world) of the
frame; one gather of the packed buckets to rank 0 ends the frame.

LibraryGather  the exchange inside the C ABI (frayhip_gather_buckets: pack, grouped RCCL send/recv peer -> root over
               xGMI, unpack); torch.distributed only carries the 128-byte communicator id to the ranks once
TileGather     the same step with torch.distributed as the transport ("gloo" in the CPU tests and one-GPU rehearsals)"""
import ctypes as C

import torch

from . import scene as _scene

BUCKET = 48


def bucket_count(W, H, first, stride):
    n = _scene.lib.frayhip_bucket_count(W, H, first, stride)
    if n < 0:
        raise ValueError("bad bucket range")
    return n


def bucket_xy(W, H, b):
    """Bucket column and row of bucket b (include/frayhip.h: rows rotated against each other by FRAYHIP_BUCKET_SKEW)."""
    bx, by = C.c_int(), C.c_int()
    if _scene.lib.frayhip_bucket_xy(W, H, b, C.byref(bx), C.byref(by)):
        raise ValueError("bad bucket")
    return bx.value, by.value


def device_pack(frame, W, H, rank, world, out, stream=None):
    rc = _scene.lib.frayhip_pack_buckets_device(frame.data_ptr(), out.data_ptr(), W, H, frame.shape[-1], rank, world, stream)
    if rc:
        raise _scene.FrayError(rc, _scene.lib.frayhip_last_error().decode())


def device_unpack(packed, frame, W, H, rank, world, stream=None):
    rc = _scene.lib.frayhip_unpack_buckets_device(packed.data_ptr(), frame.data_ptr(), W, H, frame.shape[-1], rank, world, stream)
    if rc:
        raise _scene.FrayError(rc, _scene.lib.frayhip_last_error().decode())


def _current_stream(frame):
    return torch.cuda.current_stream(frame.device).cuda_stream if frame.is_cuda else None


class LibraryGather:
    """frayhip_comm_* / frayhip_gather_buckets.  `dist` must be initialised (any backend): it broadcasts rank 0's id."""

    def __init__(self, rank, world, dist, dst=0):
        lib = _scene.lib
        self.rank, self.world, self.dst = rank, world, dst
        # ncclCommInitRank blocks until every rank has entered it: a rank that cannot bind RCCL must make ALL ranks give up here, before any
        # of them goes in (the caller then switches every rank to the other transport together)
        have = [bool(lib.frayhip_comm_available())]
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, have[0])
            if not all(every):
                raise _scene.FrayError(-4, "RCCL cannot be bound on rank(s) %s" % [r for r, ok in enumerate(every) if not ok])
        elif not have[0]:
            raise _scene.FrayError(-4, "RCCL cannot be bound in this process")
        ident = [None]
        if rank == 0:
            buf = (C.c_char * 128)()
            rc = lib.frayhip_comm_unique_id(buf)
            ident = [bytes(buf)] if rc == 0 else [(rc, lib.frayhip_last_error().decode())]
        if world > 1:
            dist.broadcast_object_list(ident, src=0)       # every rank learns rank 0's outcome: nobody is left waiting for an id that never comes
        if not isinstance(ident[0], bytes):
            raise _scene.FrayError(*ident[0])
        self._comm = C.c_void_p()
        rc = lib.frayhip_comm_create(ident[0], rank, world, C.byref(self._comm))
        if rc:
            raise _scene.FrayError(rc, lib.frayhip_last_error().decode())

    def gather(self, frame):
        H, W, ch = frame.shape
        rc = _scene.lib.frayhip_gather_buckets(self._comm, frame.data_ptr(), W, H, ch, self.dst, _current_stream(frame))
        if rc:
            raise _scene.FrayError(rc, _scene.lib.frayhip_last_error().decode())
        return frame

    def ranks_seen_by_rccl(self):
        """ncclCommCount of the communicator the gathers run on (frayhip_comm_ranks)."""
        n = _scene.lib.frayhip_comm_ranks(self._comm)
        if n < 0:
            raise _scene.FrayError(n, _scene.lib.frayhip_last_error().decode())
        return n

    def close(self):
        if self._comm:
            _scene.lib.frayhip_comm_destroy(self._comm)
            self._comm = C.c_void_p()


class TileGather:
    """Buffers and the one exchange step.  pack(frame, W, H, rank, world, out) and
    unpack(packed, frame, W, H, rank, world) default to the HIP kernels."""

    def __init__(self, W, H, channels, rank, world, device, dist, pack=device_pack, unpack=device_unpack, dst=0, stage_host=False):
        self.W, self.H, self.rank, self.world, self.dist, self.dst = W, H, rank, world, dist, dst
        self.pack, self.unpack = pack, unpack
        self.stage_host = stage_host      # transport cannot move device memory (gloo rehearsal): bounce through the host
        n = bucket_count(W, H, 0, world) * BUCKET * BUCKET * channels        # rank 0 owns the most buckets
        self.packed = torch.zeros(n, dtype=torch.float32, device=device)
        # the root's receive side: one block per rank, each the size of ONE rank's share (1 / world of the frame), i.e. one frame in total
        self.recv = [torch.zeros_like(self.packed) for _ in range(world)] if rank == dst else None

    def gather(self, frame):
        """frame: [H, W, C] float32 with this rank's buckets rendered; on rank dst it is complete on return."""
        if self.world == 1:
            return frame
        stream = _current_stream(frame)          # the stream the frame was rendered on, not the NULL stream
        kw = {"stream": stream} if self.pack is device_pack else {}
        self.pack(frame, self.W, self.H, self.rank, self.world, self.packed, **kw)
        if self.stage_host:
            host = self.packed.cpu()
            got = [torch.zeros_like(host) for _ in range(self.world)] if self.rank == self.dst else None
            self.dist.gather(host, got, dst=self.dst)
            if self.rank == self.dst:
                for r in range(self.world):
                    self.recv[r].copy_(got[r])
        else:
            self.dist.gather(self.packed, self.recv, dst=self.dst)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst:
                    self.unpack(self.recv[r], frame, self.W, self.H, r, self.world, **({"stream": stream} if self.unpack is device_unpack else {}))
        return frame
