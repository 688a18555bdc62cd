"""One process per GPU: pixel sharding and the RGBA tile gather (torch.distributed; backend "nccl" is RCCL on
ROCm, over xGMI inside a node).

The reference shards pixels over its worker pool with `(x ^ y) % threads` and returns every worker's tile to the
main process through the pool's pipes (reference data.py:70-77, init.py:185-190, 205).  Here rank g of G renders
`settings.pixels[g]` of a G-thread partition on its own GPU (scene replicated, no exchange during the march) and
one gather of the compact per-rank RGBA buffers to rank 0 replaces the pipe; rank 0 scatters them to image order.
"""
import numpy as np


def rank_pixels(width, height, world, rank):
    """[n, 2] int32 (x, y) of rank's pixels, x-major like the reference's settings.pixels[rank]."""
    x, y = np.meshgrid(np.arange(width, dtype=np.int32), np.arange(height, dtype=np.int32), indexing="ij")
    xy = np.stack([x.ravel(), y.ravel()], 1)
    return np.ascontiguousarray(xy[((xy[:, 0] ^ xy[:, 1]) % world) == rank])


def rank_pixel_counts(width, height, world):
    x, y = np.meshgrid(np.arange(width, dtype=np.int32), np.arange(height, dtype=np.int32), indexing="ij")
    return np.bincount(((x ^ y) % world).ravel(), minlength=world)


class TileGather:
    """Reusable buffers for gathering [n_px_rank, C] tiles to `dst` and scattering them into an [H, W, C] image."""

    def __init__(self, width, height, channels, dtype, device, group=None, dst=0):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.width, self.height, self.channels = width, height, channels
        counts = rank_pixel_counts(width, height, self.world)
        self.counts = [int(c) for c in counts]
        self.cap = int(counts.max())
        self.send = torch.zeros((self.cap, channels), dtype=dtype, device=device)
        self.recv = None
        self.index = None
        self.image = None
        if self.rank == dst:
            self.recv = [torch.zeros((self.cap, channels), dtype=dtype, device=device) for _ in range(self.world)]
            self.index = []
            for r in range(self.world):
                p = rank_pixels(width, height, self.world, r).astype(np.int64)
                self.index.append(torch.from_numpy(p[:, 1] * width + p[:, 0]).to(device))
            self.image = torch.zeros((height * width, channels), dtype=dtype, device=device)

    def __call__(self, local):
        """local: [counts[rank], C] tensor.  Returns the [H, W, C] image on dst, None elsewhere."""
        n = self.counts[self.rank]
        assert local.shape[0] == n
        self.send[:n].copy_(local)
        self.dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        for r in range(self.world):
            self.image[self.index[r]] = self.recv[r][: self.counts[r]]
        return self.image.view(self.height, self.width, self.channels)
