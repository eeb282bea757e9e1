"""Multi-GPU tile exchange: every rank renders the 48x48 buckets b = rank (mod world) of the
frame; one gather of the packed buckets to rank 0 ends the frame.  torch.distributed is only the
transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch

from . import scene as _scene

BUCKET = 48


def bucket_count(W, H, first, stride):
    n = _scene.lib.frayhip_bucket_count(W, H, first, stride)
    if n < 0:
        raise ValueError("bad bucket range")
    return n


def device_pack(frame, W, H, rank, world, out, stream=None):
    rc = _scene.lib.frayhip_pack_buckets_device(frame.data_ptr(), out.data_ptr(), W, H, frame.shape[-1], rank, world, stream)
    if rc:
        raise _scene.FrayError(rc, _scene.lib.frayhip_last_error().decode())


def device_unpack(packed, frame, W, H, rank, world, stream=None):
    rc = _scene.lib.frayhip_unpack_buckets_device(packed.data_ptr(), frame.data_ptr(), W, H, frame.shape[-1], rank, world, stream)
    if rc:
        raise _scene.FrayError(rc, _scene.lib.frayhip_last_error().decode())


class TileGather:
    """Buffers and the one exchange step.  pack(frame, W, H, rank, world, out) and
    unpack(packed, frame, W, H, rank, world) default to the HIP kernels."""

    def __init__(self, W, H, channels, rank, world, device, dist, pack=device_pack, unpack=device_unpack, dst=0, stage_host=False):
        self.W, self.H, self.rank, self.world, self.dist, self.dst = W, H, rank, world, dist, dst
        self.pack, self.unpack = pack, unpack
        self.stage_host = stage_host      # transport cannot move device memory (gloo rehearsal): bounce through the host
        n = bucket_count(W, H, 0, world) * BUCKET * BUCKET * channels        # rank 0 owns the most buckets
        self.packed = torch.zeros(n, dtype=torch.float32, device=device)
        self.recv = [torch.zeros_like(self.packed) for _ in range(world)] if rank == dst else None

    def gather(self, frame):
        """frame: [H, W, C] float32 with this rank's buckets rendered; on rank dst it is complete on return."""
        if self.world == 1:
            return frame
        self.pack(frame, self.W, self.H, self.rank, self.world, self.packed)
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
                    self.unpack(self.recv[r], frame, self.W, self.H, r, self.world)
        return frame
