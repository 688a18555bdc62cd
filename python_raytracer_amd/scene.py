"""Flattening of Camera.chunks into the packed voxel / material SoA the HIP kernels read (include/vrt.h,
`vrt_scene`).  This is the `chunks -> packed SoA` step of the drop-in boundary; it replaces the
per-lookup dict / box scan of the reference's Frame.get_voxel (reference data.py:136-145).
"""
import numpy as np

from .lib import is_default_material_function

MAT_PROPS = ("roughness", "absorption", "ior", "energy")


def material_row(m):
    """[r, g, b, roughness, absorption, ior, energy, 0] from a Material-like object; validates what the on-device
    default shader needs (reference lib.py:448-460 reads exactly these)."""
    f = getattr(m, "function", None)
    if not is_default_material_function(f):
        raise TypeError("Material.function must be the default shader lib.material: custom per-ray Python "
                        "callbacks cannot run inside the GPU kernel (got %r)" % (f,))
    try:
        alb = m.albedo
        row = [float(alb.r), float(alb.g), float(alb.b)] + [float(getattr(m, k)) for k in MAT_PROPS] + [0.0]
    except AttributeError as e:
        raise TypeError("Material is missing a render property: %s" % e) from None
    if not (row[4] >= 0):
        raise ValueError("Material.absorption must be >= 0 (got %r)" % row[4])
    return row


def pack_blocks(blocks):
    """[n, cs, cs, cs] uint8 (x, y, z) -> [n, cs^3] bytes in the bricked order of vrt_voxel_offset():
    8^3 bricks [bx][by][bz], 4^3 micro-bricks [mx][my][mz], voxels [x][y][z]."""
    n, cs = blocks.shape[0], blocks.shape[1]
    nb = cs // 8
    b = blocks.reshape(n, nb, 2, 4, nb, 2, 4, nb, 2, 4)
    b = b.transpose(0, 1, 4, 7, 2, 5, 8, 3, 6, 9)
    return np.ascontiguousarray(b).reshape(n, cs * cs * cs)


def unpack_blocks(packed, cs):
    """Inverse of pack_blocks: [n, cs^3] bytes -> [n, cs, cs, cs] (x, y, z)."""
    n, nb = packed.shape[0], cs // 8
    b = np.asarray(packed).reshape(n, nb, nb, nb, 2, 2, 2, 4, 4, 4).transpose(0, 1, 4, 7, 2, 5, 8, 3, 6, 9)
    return np.ascontiguousarray(b).reshape(n, cs, cs, cs)


class PackedScene:
    """Host arrays of a flattened scene; `.to(device)` uploads them as torch tensors."""

    def __init__(self, origin, dims, chunk_size, chunk_table, voxels, materials):
        self.origin = np.asarray(origin, np.int64)
        self.dims = np.asarray(dims, np.int32)
        self.chunk_size = int(chunk_size)
        self.chunk_table = np.ascontiguousarray(chunk_table, np.uint32).reshape(-1)
        self.voxels = np.ascontiguousarray(voxels, np.uint8)
        self.materials = np.ascontiguousarray(materials, np.float64).reshape(-1, 8)
        self.n_slots = int(self.voxels.shape[0]) if self.voxels.ndim == 2 else 0
        # largest Frame.resolution in the table (selects the kernel variant only; 0 = unknown)
        res = self.chunk_table >> 24
        self.max_resolution = int(res.max()) if self.chunk_table.size and self.n_slots else 1
        self.device_tensors = None
        self._table_identity = None

    def table_identity(self):
        """VRT_SCENE_TABLE_IS_IDENTITY (include/vrt.h): every cell of the chunk table holds (index + 1) | 1 << 24 -- a dense
        world at resolution 1 whose voxel blocks lie in table order.  Checked once, on the table itself (host or
        device); VRT_TABLE_IDENTITY=0 never reports it (the march then reads the table: same results)."""
        import os
        if os.environ.get("VRT_TABLE_IDENTITY", "1") in ("0", ""):
            return False
        if self._table_identity is None:
            cells = int(np.prod(self.dims))
            ok = cells == self.n_slots and cells > 0
            if ok:
                t = (self.device_tensors or {}).get("chunk_table") if getattr(self, "resident", False) else None
                if t is not None:
                    import torch
                    want = (torch.arange(1, cells + 1, dtype=torch.int32, device=t.device) | (1 << 24))
                    ok = t.numel() == cells and bool((t.view(torch.int32).reshape(-1) == want).all())
                else:
                    ok = self.chunk_table.size == cells and bool(
                        (self.chunk_table == ((np.arange(1, cells + 1, dtype=np.uint32)) | np.uint32(1 << 24))).all())
            self._table_identity = bool(ok)
        return self._table_identity

    @staticmethod
    def check_chunk_size(cs):
        if cs < 8 or cs & (cs - 1) or cs > 256:
            raise ValueError("chunk_size must be a power of two in [8, 256] for the GPU path (got %r)" % (cs,))

    @classmethod
    def from_dense(cls, origin, dims, chunk_size, present, res, grid, materials):
        """grid: [X, Y, Z] uint8 material ids at world coordinates (origin + index); present/res: [dims] uint8.
        materials: [n, 7] rows (r, g, b, roughness, absorption, ior, energy)."""
        cs = int(chunk_size)
        cls.check_chunk_size(cs)
        dims = np.asarray(dims, np.int64)
        present = np.asarray(present, np.uint8).reshape(tuple(dims))
        res = np.asarray(res, np.uint8).reshape(tuple(dims))
        grid = np.asarray(grid, np.uint8)
        assert grid.shape == tuple(dims * cs), (grid.shape, dims, cs)
        if np.any(np.asarray(origin) % cs):
            raise ValueError("scene origin must be a multiple of chunk_size")
        if present.any() and res[present > 0].min() < 1:
            raise ValueError("Frame.resolution must be >= 1")
        blocks = grid.reshape(dims[0], cs, dims[1], cs, dims[2], cs).transpose(0, 2, 4, 1, 3, 5)
        sel = present > 0
        vox = pack_blocks(np.ascontiguousarray(blocks[sel]))
        table = np.zeros(tuple(dims), np.uint32)
        table[sel] = (np.arange(int(sel.sum()), dtype=np.uint32) + 1) | (res[sel].astype(np.uint32) << 24)
        mats = np.zeros((len(materials), 8), np.float64)
        mats[:, :7] = np.asarray(materials, np.float64).reshape(-1, 7)
        if len(mats) > 255:
            raise ValueError("at most 255 materials (u8 voxel ids)")
        if grid.max(initial=0) > len(mats):
            raise ValueError("voxel id exceeds the material table")
        return cls(origin, dims, cs, table, vox, mats)

    @classmethod
    def from_chunks(cls, chunks, chunk_size):
        """chunks: {(x, y, z) chunk position: Frame-like with .resolution and data3/data6 (or .cells())}.
        Returns (PackedScene, [materials in id order])."""
        cs = int(chunk_size)
        cls.check_chunk_size(cs)
        mats, mat_id, rows = [], {}, []
        if not chunks:
            sc = cls([0, 0, 0], [1, 1, 1], cs, np.zeros(1, np.uint32), np.zeros((0, cs ** 3), np.uint8),
                     np.zeros((0, 8)))
            return sc, mats
        keys = np.array(list(chunks.keys()), np.float64)
        if np.any(keys != np.floor(keys)) or np.any(keys.astype(np.int64) % cs):
            raise ValueError("chunk positions must be integer multiples of chunk_size")
        keys = keys.astype(np.int64)
        lo = keys.min(0)
        dims = (keys.max(0) - lo) // cs + 1
        table = np.zeros(tuple(dims), np.uint32)
        blocks = np.zeros((len(chunks), cs, cs, cs), np.uint8)
        for slot, (post, fr) in enumerate(chunks.items()):
            post = np.array([int(v) for v in post], np.int64)
            r = int(fr.resolution)
            if r < 1 or r > 255:
                raise ValueError("Frame.resolution must be in [1, 255]")
            cells = fr.cells() if hasattr(fr, "cells") else _cells_of(fr)
            blk = blocks[slot]
            for q, m in cells:
                i = mat_id.get(id(m))
                if i is None:
                    rows.append(material_row(m))
                    mats.append(m)
                    if len(mats) > 255:
                        raise ValueError("at most 255 distinct materials (u8 voxel ids)")
                    i = mat_id[id(m)] = len(mats)
                lx, ly, lz = q[0] * r - post[0], q[1] * r - post[1], q[2] * r - post[2]
                if not (0 <= lx < cs and 0 <= ly < cs and 0 <= lz < cs):
                    raise ValueError("Frame of chunk %s holds a voxel at world %s outside the chunk; camera chunk "
                                     "Frames must stay inside [pos, pos + chunk_size)"
                                     % (tuple(post), (q[0] * r, q[1] * r, q[2] * r)))
                blk[lx, ly, lz] = i
            c = (post - lo) // cs
            table[tuple(c)] = (slot + 1) | (r << 24)
        sc = cls(lo, dims, cs, table, pack_blocks(blocks), np.array(rows, np.float64).reshape(-1, 8))
        return sc, mats

    @classmethod
    def from_device(cls, origin, dims, chunk_size, chunk_table, voxels, n_slots, materials, max_resolution=0,
                    occupancy=None):
        """Wrap voxel data that already lives on the device (torch tensors): chunk_table int32 [prod(dims)],
        voxels uint8 [n_slots * chunk_size^3] in the packed order; materials: host [n, 7] rows.  max_resolution:
        largest resolution in chunk_table if the caller knows it (vrt_voxelize / vrt_synth_volume write 1), 0 = unknown
        (the generic-resolution kernel).  It must not be understated: the resolution-1 and resolution <= 2 kernels leave
        out the snapping a larger resolution needs."""
        import torch
        mats = np.zeros((len(materials), 8), np.float64)
        mats[:, :7] = np.asarray(materials, np.float64).reshape(-1, 7)
        sc = cls(origin, dims, chunk_size, np.zeros(1, np.uint32), np.zeros((0, int(chunk_size) ** 3), np.uint8), mats)
        sc.n_slots = int(n_slots)
        sc.max_resolution = int(max_resolution)
        sc.device_tensors = dict(chunk_table=chunk_table, voxels=voxels,
                                 occupancy=build_occupancy(voxels, int(n_slots) * int(chunk_size) ** 3, occupancy)
                                 if want_occupancy() else None,
                                 materials=torch.from_numpy(mats.reshape(-1)).to(voxels.device) if mats.size else
                                 torch.zeros(8, dtype=torch.float64, device=voxels.device))
        sc.resident = True
        return sc

    def to(self, device):
        import torch
        if getattr(self, "resident", False):
            return self
        self.device_tensors = dict(
            chunk_table=torch.from_numpy(self.chunk_table.view(np.int32)).to(device),
            voxels=torch.from_numpy(self.voxels.reshape(-1)).to(device) if self.voxels.size else
            torch.zeros(1, dtype=torch.uint8, device=device),
            materials=torch.from_numpy(self.materials.reshape(-1)).to(device) if self.materials.size else
            torch.zeros(8, dtype=torch.float64, device=device),
        )
        self.device_tensors["occupancy"] = build_occupancy(self.device_tensors["voxels"],
                                                           self.n_slots * self.chunk_size ** 3) if want_occupancy() else None
        return self


def want_occupancy():
    """The occupancy words are only read by the measurement variants of the march (VRT_LOOKUP=1|2, vrt_kernels.hip)."""
    import os
    return os.environ.get("VRT_LOOKUP", "0") in ("1", "2")


def build_occupancy(voxels, n_bytes, out=None):
    """Occupancy words of a packed voxel buffer on the device (vrt_occupancy_build): bit b of word w says whether
    voxels[64 w + b] holds a material -- one word per 4^3 micro-brick."""
    import torch
    from . import _native as nat
    words = max(1, n_bytes // 64)
    if out is None or out.numel() < words:
        out = torch.zeros(words, dtype=torch.int64, device=voxels.device)
    if n_bytes:
        with torch.cuda.device(voxels.device):
            nat.check(nat.lib().vrt_occupancy_build(voxels.data_ptr(), n_bytes, out.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "vrt_occupancy_build")
    return out


def _cells_of(fr):
    yield from fr.data3.items()
    for b, mat in fr.data6.items():
        for x in range(b[0], b[3] + 1):
            for y in range(b[1], b[4] + 1):
                for z in range(b[2], b[5] + 1):
                    yield (x, y, z), mat
