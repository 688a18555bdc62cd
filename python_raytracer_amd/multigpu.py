"""One process per GPU: pixel sharding and the RGBA tile gather (torch.distributed; backend "nccl" is RCCL on
ROCm, over xGMI inside a node).

The reference shards pixels over its worker pool with `(x ^ y) % threads` and returns every worker's tile to the
main process through the pool's pipes (reference data.py:70-77, init.py:185-190, 205).  Here rank g of G renders
`settings.pixels[g]` of a G-thread partition on its own GPU (scene replicated, no exchange during the march) and
one gather of the compact per-rank RGBA buffers to rank 0 replaces the pipe; rank 0 scatters them to image order.
"""
import functools

import numpy as np


PARTITIONS = ("xor", "seed", "tiles")
TILE = 8  # block edge of the "tiles" partition


def _primes_upto(n):
    return [p for p in range(2, n + 1) if all(p % q for q in range(2, int(p ** 0.5) + 1))]


@functools.lru_cache(maxsize=4)
def owner_map(width, height, world, partition="xor", samples=1):
    """[width, height] int32: the rank that renders pixel (x, y).

    "xor": (x ^ y) % world, the reference's worker partition (reference data.py:74-77).
    "seed": pixels grouped by static-seed class.  A static-mode ray of pixel (x, y), sample s is seeded with
    (1 + x)(1 + y)(1 + s) (reference init.py:30-31) and the draw table is built once per DISTINCT seed, so a rank
    whose pixels share seeds only among themselves seeds 1/world of the frame's distinct seeds instead of ~3/world
    (config 3, 8 ranks: 1.04 M instead of 3.09 M seeds per rank).  Two pixels can only share a seed if
    (1 + x)(1 + y) agree after removing every prime factor <= samples, so whole classes of that reduced product are
    dealt to the ranks, largest first to the least-loaded rank, which also balances the pixel counts
    to a fraction of a percent.
    "tiles": TILE x TILE pixel blocks dealt to the ranks along the diagonals, ((x // TILE) + (y // TILE)) % world: a
    rank's list (x-major) then holds runs of TILE vertically adjacent pixels, like the single-GPU frame, while the
    blocks of every rank are spread over the whole image.  Measured on MI355X (tools/exp_share.py, a 1/8 share of
    config 3): the march takes 1.05 ms with "tiles", 1.06 with "seed", 1.07 with "xor" -- the partition does not
    decide the share's cost; its fixed part (about 0.25 ms of draining waves per frame) does.
    The image is the same either way: which rank renders a pixel never changes its colour."""
    x, y = np.meshgrid(np.arange(width, dtype=np.int64), np.arange(height, dtype=np.int64), indexing="ij")
    if partition == "xor" or world == 1:
        return ((x ^ y) % world).astype(np.int32)
    if partition == "tiles":
        return (((x // TILE) + (y // TILE)) % world).astype(np.int32)
    if partition != "seed":
        raise ValueError("partition must be one of %r" % (PARTITIONS,))
    def reduced(n):  # 1..n with the small prime factors removed; the reduction is multiplicative
        v = np.arange(1, n + 1, dtype=np.int64)
        for p in _primes_upto(max(2, int(samples))):
            while True:
                m = (v % p) == 0
                if not m.any():
                    break
                v[m] //= p
        return v
    v = (reduced(width)[:, None] * reduced(height)[None, :]).ravel()
    classes, inverse, counts = np.unique(v, return_inverse=True, return_counts=True)
    order = np.argsort(-counts, kind="stable")
    k = np.arange(len(classes)) % (2 * world)
    owner_sorted = np.where(k < world, k, 2 * world - 1 - k).astype(np.int32)
    load = np.zeros(world, np.int64)
    head = min(len(classes), 4096)  # the few big classes: least-loaded rank first
    for i in range(head):
        r = int(np.argmin(load))
        owner_sorted[i] = r
        load[r] += counts[order[i]]
    if head < len(classes):  # the many small ones: snake order, starting from the least-loaded rank
        owner_sorted[head:] = np.argsort(load, kind="stable").astype(np.int32)[owner_sorted[head:]]
    owner_of_class = np.empty(len(classes), np.int32)
    owner_of_class[order] = owner_sorted
    return owner_of_class[inverse].reshape(width, height)


def rank_pixels(width, height, world, rank, partition="xor", samples=1):
    """[n, 2] int32 (x, y) of rank's pixels, x-major like the reference's settings.pixels[rank]."""
    own = owner_map(width, height, world, partition, samples)
    xs, ys = np.nonzero(own == rank)
    return np.ascontiguousarray(np.stack([xs, ys], 1).astype(np.int32))


def rank_pixel_counts(width, height, world, partition="xor", samples=1):
    return np.bincount(owner_map(width, height, world, partition, samples).ravel(), minlength=world)


_SIGN = -(1 << 63)


def union_traversed(keys, group=None):
    """Cross-rank union of the traversed-chunk feedback (reference init.py:189 keeps one list per worker and
    init.py:393 culls against `unpack(self.traversed)`, the union of all of them).  `keys` is the int64 visit-key
    tensor of a RenderResult (UINT64_MAX = never visited, else ray_index << 12 | resnap_index); every rank renders
    with the same camera and therefore the same box, so one in-place MIN all-reduce over the keys read as UNSIGNED
    (sign bit flipped around the reduce) leaves on every rank: visited by any rank <=> key != UINT64_MAX.  The
    order of RenderResult.traversed() afterwards is by the smallest rank-local key; membership, which is all the
    culling loop reads (init.py:447), is exact.  No-op for a single process."""
    import torch.distributed as dist
    if keys is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return keys
    keys.bitwise_xor_(_SIGN)
    dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)
    keys.bitwise_xor_(_SIGN)
    return keys


def merge_traversed(key_tensors):
    """Union of several visit-key tensors over the same box (several tiles of one process, like the reference's
    per-thread lists): element-wise unsigned minimum.  Returns a new tensor."""
    import torch
    out = key_tensors[0] ^ _SIGN
    for k in key_tensors[1:]:
        out = torch.minimum(out, k ^ _SIGN)
    return out ^ _SIGN


def chunk_update_all_ranks(cam, result, group=None):
    """Camera.chunk_update with the traversed feedback of ALL ranks (culling on, reference config default): every
    rank ends up with the same camera chunk table a single-process run would select."""
    if result is not None and result.traversed_keys is not None:
        union_traversed(result.traversed_keys, group)
    return cam.chunk_update(result)


class TileGather:
    """Reusable buffers for gathering [n_px_rank, C] tiles to `dst` and scattering them into an [H, W, C] image.

    `gather(local)` is the blocking form.  `submit(local)` / `collect()` pipeline one frame deep: the gather of frame
    k runs on the communication stream while frame k + 1 renders, and collect() scatters frame k into image order."""

    def __init__(self, width, height, channels, dtype, device, group=None, dst=0, partition="xor", samples=1):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.width, self.height, self.channels = width, height, channels
        counts = rank_pixel_counts(width, height, self.world, partition, samples)
        self.counts = [int(c) for c in counts]
        self.cap = int(counts.max())
        self.send = [torch.zeros((self.cap, channels), dtype=dtype, device=device) for _ in range(2)]
        self.recv = self.index = self.image = None
        own = rank_pixels(width, height, self.world, self.rank, partition, samples).astype(np.int64)
        self.own_index = torch.from_numpy(own[:, 1] * width + own[:, 0]).to(device)  # this rank's pixels, image order
        self.pending = None
        self.frame = 0
        if self.rank == dst:
            self.recv = [torch.zeros((self.world, self.cap, channels), dtype=dtype, device=device) for _ in range(2)]
            # one scatter for all ranks: row r * cap + i of recv goes to pixel index[r * cap + i]; the padding rows
            # of the shorter tiles go to a spare row behind the image
            idx = np.full((self.world, self.cap), height * width, np.int64)
            for r in range(self.world):
                p = rank_pixels(width, height, self.world, r, partition, samples).astype(np.int64)
                idx[r, :len(p)] = p[:, 1] * width + p[:, 0]
            self.index = torch.from_numpy(idx.ravel()).to(device)
            self.image = torch.zeros((height * width + 1, channels), dtype=dtype, device=device)

    def submit(self, local=None, window=None):
        """Start gathering this rank's tile: `local` ([counts[rank], C], pixel-list order) or `window` (the rank's
        full [H, W, C] window image as Camera.tile paints it, from which its own pixels are picked).  At most one
        gather is in flight."""
        assert self.pending is None, "collect() the previous frame first"
        n = self.counts[self.rank]
        b = self.frame & 1
        self.frame += 1
        if window is not None:
            self.torch.index_select(window.reshape(self.height * self.width, self.channels), 0, self.own_index,
                                    out=self.send[b][:n])
        else:
            assert local.shape[0] == n
            self.send[b][:n].copy_(local)
        out = [self.recv[b][r] for r in range(self.world)] if self.rank == self.dst else None
        work = self.dist.gather(self.send[b], out, dst=self.dst, group=self.group, async_op=True)
        self.pending = (work, b)

    def collect(self):
        """Finish the gather in flight.  Returns the [H, W, C] image on dst, None elsewhere."""
        work, b = self.pending
        self.pending = None
        work.wait()
        if self.rank != self.dst:
            return None
        self.image[self.index] = self.recv[b].view(self.world * self.cap, self.channels)
        return self.image[: self.height * self.width].view(self.height, self.width, self.channels)

    def __call__(self, local):
        self.submit(local)
        return self.collect()
