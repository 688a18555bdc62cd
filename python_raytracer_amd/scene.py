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


DENSE_SMALL_BYTES = 64 << 20     # scene boxes up to this many voxel bytes are always laid out in table order
DENSE_MAX_BYTES = 1 << 30        # ... larger ones when at least half of their cells hold a Frame; never beyond this


def dense_layout(dims, chunk_size, n_present):
    """Should a scene of this box get one voxel block per CELL, in table order (VRT_SCENE_LAYOUT_DENSE, include/vrt.h)?
    The march can then look ahead across chunk borders (world-axis offset tables).  VRT_DENSE=0 keeps one block per
    present chunk (the march then stops its look-ahead at chunk borders: same results)."""
    import os
    if os.environ.get("VRT_DENSE", "1") in ("0", ""):
        return False
    cells = int(np.prod(np.asarray(dims, np.int64)))
    nbytes = cells * int(chunk_size) ** 3
    return cells < (1 << 24) and (nbytes <= DENSE_SMALL_BYTES or (nbytes <= DENSE_MAX_BYTES and cells <= 2 * int(n_present)))


def build_world_tables(dims, chunk_size, device):
    """vrt_world_tables_build: the world-axis offset tables of a table-order scene on `device`, or None when the march
    could not use them (include/vrt.h)."""
    import ctypes as C
    import torch
    from . import _native as nat
    L = nat.lib()
    d32 = (C.c_int32 * 3)(*[int(v) for v in dims])
    nb = C.c_int64(0)
    nat.check(L.vrt_world_tables_bytes(d32, int(chunk_size), C.byref(nb)), "vrt_world_tables_bytes")
    if nb.value == 0:
        return None
    out = torch.empty(nb.value // 4, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        nat.check(L.vrt_world_tables_build(d32, int(chunk_size), out.data_ptr(), nb.value,
                                           torch.cuda.current_stream().cuda_stream), "vrt_world_tables_build")
    return out


class PackedScene:
    """Host arrays of a flattened scene; `.to(device)` uploads them as torch tensors."""

    def __init__(self, origin, dims, chunk_size, chunk_table, voxels, materials, dense=False):
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
        # the voxel blocks lie in table order: block i belongs to cell i, listed in the table or not (VRT_SCENE_LAYOUT_DENSE)
        self.dense = bool(dense)

    def table_identity(self):
        """VRT_SCENE_TABLE_IS_IDENTITY (include/vrt.h): every cell of the chunk table holds (index + 1) | 1 << 24 -- a dense
        world at resolution 1 whose voxel blocks lie in table order.  Checked once, on the table itself (host or
        device); VRT_TABLE_IDENTITY=0 never reports it (the march then reads the table: same results)."""
        import os
        if os.environ.get("VRT_TABLE_IDENTITY", "1") in ("0", ""):
            return False
        if self._table_identity is None:
            # (host arrays: checked here, once.  Device-resident scenes settle it where their table is produced --
            # from_device -- never on the frame path, and a table that is rewritten in place never claims it.)
            cells = int(np.prod(self.dims))
            ok = cells == self.n_slots and cells > 0 and not getattr(self, "resident", False)
            if ok:
                ok = self.chunk_table.size == cells and bool(
                    (self.chunk_table == ((np.arange(1, cells + 1, dtype=np.uint32)) | np.uint32(1 << 24))).all())
            self._table_identity = bool(ok)
        return self._table_identity

    def layout_flags(self):
        """vrt_scene.flags of this scene."""
        from . import _native as nat
        return (nat.SCENE_TABLE_IS_IDENTITY if self.table_identity() else 0) | (nat.SCENE_LAYOUT_DENSE if self.dense else 0)

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
        dense = dense_layout(dims, cs, int(sel.sum()))
        table = np.zeros(tuple(dims), np.uint32)
        if dense:   # one block per cell, in table order (the blocks of cells without a Frame are never interpreted)
            vox = pack_blocks(np.ascontiguousarray(blocks).reshape(-1, cs, cs, cs))
            table[sel] = (np.flatnonzero(sel.reshape(-1)).astype(np.uint32) + 1) | (res[sel].astype(np.uint32) << 24)
        else:
            vox = pack_blocks(np.ascontiguousarray(blocks[sel]))
            table[sel] = (np.arange(int(sel.sum()), dtype=np.uint32) + 1) | (res[sel].astype(np.uint32) << 24)
        mats = np.zeros((len(materials), 8), np.float64)
        mats[:, :7] = np.asarray(materials, np.float64).reshape(-1, 7)
        if len(mats) > 255:
            raise ValueError("at most 255 materials (u8 voxel ids)")
        if grid.max(initial=0) > len(mats):
            raise ValueError("voxel id exceeds the material table")
        return cls(origin, dims, cs, table, vox, mats, dense=dense)

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
        dense = dense_layout(dims, cs, len(chunks))
        blocks = np.zeros((int(np.prod(dims)) if dense else len(chunks), cs, cs, cs), np.uint8)
        for slot, (post, fr) in enumerate(chunks.items()):
            post = np.array([int(v) for v in post], np.int64)
            if dense:   # block = cell
                c = (post - lo) // cs
                slot = int((c[0] * dims[1] + c[1]) * dims[2] + c[2])
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
        sc = cls(lo, dims, cs, table, pack_blocks(blocks), np.array(rows, np.float64).reshape(-1, 8), dense=dense)
        return sc, mats

    @classmethod
    def from_device(cls, origin, dims, chunk_size, chunk_table, voxels, n_slots, materials, max_resolution=0,
                    occupancy=None, dense=None, table_identity=None):
        """Wrap voxel data that already lives on the device (torch tensors): chunk_table int32 [prod(dims)],
        voxels uint8 [n_slots * chunk_size^3] in the packed order; materials: host [n, 7] rows.  max_resolution:
        largest resolution in chunk_table if the caller knows it (vrt_voxelize / vrt_synth_volume write 1), 0 = unknown
        (the generic-resolution kernel).  It must not be understated: the resolution-1 and resolution <= 2 kernels leave
        out the snapping a larger resolution needs.
        dense: the blocks lie in table order (VRT_SCENE_LAYOUT_DENSE: what vrt_voxelize / vrt_synth_volume write);
        None = checked here, on the device.  table_identity: every cell holds (index + 1) | 1 << 24
        (VRT_SCENE_TABLE_IS_IDENTITY); None = checked here, on the device.  Both checks look at the table as it is NOW: a
        caller that rewrites it in place later (DeviceWorld) passes table_identity=False and vouches for `dense`."""
        import torch
        mats = np.zeros((len(materials), 8), np.float64)
        mats[:, :7] = np.asarray(materials, np.float64).reshape(-1, 7)
        sc = cls(origin, dims, chunk_size, np.zeros(1, np.uint32), np.zeros((0, int(chunk_size) ** 3), np.uint8), mats)
        sc.n_slots = int(n_slots)
        sc.max_resolution = int(max_resolution)
        cells = int(np.prod(np.asarray(dims, np.int64)))
        if cells != sc.n_slots or cells == 0 or chunk_table.numel() != cells:
            dense = table_identity = False
        t32 = chunk_table.view(torch.int32).reshape(-1)
        if table_identity is None or dense is None:
            want = torch.arange(1, cells + 1, dtype=torch.int32, device=chunk_table.device)
            if table_identity is None:
                table_identity = bool((t32 == (want | (1 << 24))).all())
            if dense is None:
                slot = t32 & 0xffffff
                dense = bool(((slot == 0) | (slot == want)).all())
        sc._table_identity = bool(table_identity)
        sc.dense = bool(dense) or bool(table_identity)
        sc.device_tensors = dict(chunk_table=chunk_table, voxels=voxels,
                                 occupancy=build_occupancy(voxels, int(n_slots) * int(chunk_size) ** 3, occupancy)
                                 if want_occupancy() else None,
                                 materials=torch.from_numpy(mats.reshape(-1)).to(voxels.device) if mats.size else
                                 torch.zeros(8, dtype=torch.float64, device=voxels.device),
                                 world_tables=build_world_tables(dims, chunk_size, voxels.device) if sc.dense else None)
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
        self.device_tensors["world_tables"] = build_world_tables(self.dims, self.chunk_size, device) \
            if (self.dense or self.table_identity()) and self.n_slots else None
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
