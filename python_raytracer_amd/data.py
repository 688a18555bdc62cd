"""Host-side scene inputs of the trace path: render settings, the pixel partition, Material and Frame.

Mirrors the parts of the reference's `data` module that Camera.tile / Camera.trace read
(reference data.py:15-77 settings + pixels, 85-93 Material, 96-145 Frame lookup).  Physics, sprites,
objects and the mod loader are out of scope.
"""
import configparser

import numpy as np

from .lib import store, material_background

# keys of the [RENDER]/[WINDOW] sections the path reads, with the reference's `value or default` idiom
_INT_KEYS = dict(width=64, height=64, samples=1, chunk_size=16, chunk_lod=0, dist_min=0, dist_max=32, threads=0)
_FLOAT_KEYS = dict(shutter=0.0, falloff=0.0, fov=90.0, dof=0.0, max_light=0.0, max_bounces=0.0, lod_bounces=0.0,
                   lod_samples=0.0, lod_random=0.0, lod_edge=0.0)
_BOOL_KEYS = dict(static=False, culling=False, sync=False)
_SECTION = dict(width="WINDOW", height="WINDOW")


class PixelList:
    """settings.pixels[t]: the (x, y) pixels of render thread t in the reference's x-major order
    (reference data.py:72-77).  Behaves like the reference's list of tuples; `.array` is the [n, 2] int32
    numpy view the GPU path uploads."""

    def __init__(self, array):
        self.array = np.ascontiguousarray(array, np.int32).reshape(-1, 2)

    def __len__(self):
        return len(self.array)

    def __iter__(self):
        return iter(map(tuple, self.array.tolist()))

    def __getitem__(self, i):
        r = self.array[i]
        return tuple(int(v) for v in r) if r.ndim == 1 else PixelList(r)

    def __eq__(self, other):
        return list(self) == list(other)


def pixel_partition(width, height, threads):
    """Pixels of thread t = {(x, y): (x ^ y) % threads == t}, enumerated x-major (reference data.py:70-77)."""
    x, y = np.meshgrid(np.arange(width, dtype=np.int32), np.arange(height, dtype=np.int32), indexing="ij")
    xy = np.stack([x.ravel(), y.ravel()], 1)
    t = (xy[:, 0] ^ xy[:, 1]) % threads
    return [PixelList(xy[t == k]) for k in range(threads)]


def finalize_settings(s):
    """Recompute the derived fields (reference data.py:64-77) after width/height/threads/chunk_size changed."""
    s.window = s.width, s.height
    s.proportions = ((s.width + s.height) / 2) / max(s.width, s.height)
    s.chunk_radius = round(s.chunk_size / 2)
    s.pixels = pixel_partition(s.width, s.height, s.threads)
    return s


def make_settings(**overrides):
    """Settings store with the default mod's [RENDER] values (reference mods/default/config.cfg:1-34)."""
    s = store(width=64, height=48, samples=1, static=True, culling=True, sync=False, shutter=0.25, falloff=0.25,
              chunk_size=16, chunk_lod=2, fov=90.0, dof=0.5, dist_min=0, dist_max=192, max_light=1.0,
              max_bounces=2.0, lod_bounces=0.5, lod_samples=0.5, lod_random=0.25, lod_edge=0.25, threads=1)
    for k, v in overrides.items():
        setattr(s, k, v)
    return finalize_settings(s)


def load_settings(path, threads=None):
    """Read a mod's config.cfg the way the reference does (data.py:15-63): `value or default`, so a configured
    0 falls back to the default; threads = 0 means one render thread per visible GPU here (the reference uses
    the CPU count)."""
    cfg = configparser.RawConfigParser()
    if not cfg.read(path):
        raise FileNotFoundError(path)
    s = store()
    for k, d in _INT_KEYS.items():
        setattr(s, k, cfg.getint(_SECTION.get(k, "RENDER"), k) or d)
    for k, d in _FLOAT_KEYS.items():
        setattr(s, k, cfg.getfloat("RENDER", k) or d)
    for k, d in _BOOL_KEYS.items():
        setattr(s, k, cfg.getboolean("RENDER", k) or d)
    if threads is not None:
        s.threads = threads
    if not s.threads:
        s.threads = max(1, _gpu_count())
    return finalize_settings(s)


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 1


# module globals, as in the reference (data.py:18, 80-82)
settings = make_settings()
background = material_background


class Material:
    """Attribute bag describing one voxel material (reference data.py:85-93).  The trace path reads
    `function, albedo, roughness, absorption, ior, energy`; other attributes are kept but ignored."""

    def __init__(self, **props):
        self.function = props.get("function")
        for k, v in props.items():
            setattr(self, k, v)

    def copy(self):
        import copy
        return copy.deepcopy(self)


class Frame:
    """Sparse voxel store of one chunk (reference data.py:96-145): `data3` maps a cell (x, y, z) to a Material,
    `data6` maps an inclusive box (x0, y0, z0, x1, y1, z1) to a Material, cells are world coordinates divided
    by `resolution`.  Box packing (reference data.py:192-250) is lossless and is not performed here; Frames
    that carry `data6` boxes (e.g. built by the reference) are read correctly."""

    def __init__(self, **kw):
        self.packed = kw.get("packed", False)
        self.resolution = kw.get("resolution", 1)
        self.data3 = {}
        self.data6 = {}

    def clear(self):
        self.data3 = {}
        self.data6 = {}

    def _cell(self, pos):
        p = pos.tuple() if hasattr(pos, "tuple") else tuple(pos)
        if self.resolution > 1:
            p = tuple(v // self.resolution for v in p)
        return p

    def get_voxel(self, pos):
        """Material at world position `pos` (integers) or None (reference data.py:136-145)."""
        q = self._cell(pos)
        m = self.data3.get(q)
        if m is not None:
            return m
        for b, mat in self.data6.items():
            if b[0] <= q[0] <= b[3] and b[1] <= q[1] <= b[4] and b[2] <= q[2] <= b[5]:
                return mat
        return None

    def set_voxels(self, voxels, force=True):
        """voxels: {(x, y, z) world position: Material or None}; positions not divisible by the resolution are
        skipped (reference data.py:163-175)."""
        r = self.resolution
        for post, mat in voxels.items():
            if r > 1 and (post[0] % r or post[1] % r or post[2] % r):
                continue
            if not force and self.get_voxel(post):
                continue
            q = tuple(v // r for v in post) if r > 1 else tuple(post)
            self._unbox(q)
            if mat:
                self.data3[q] = mat
            else:
                self.data3.pop(q, None)

    def set_voxel(self, pos, mat, force=True):
        self.set_voxels({(pos.tuple() if hasattr(pos, "tuple") else tuple(pos)): mat}, force)

    def _unbox(self, q):
        for b, mat in list(self.data6.items()):
            if b[0] <= q[0] <= b[3] and b[1] <= q[1] <= b[4] and b[2] <= q[2] <= b[5]:
                for x in range(b[0], b[3] + 1):
                    for y in range(b[1], b[4] + 1):
                        for z in range(b[2], b[5] + 1):
                            self.data3[(x, y, z)] = mat
                del self.data6[b]
                return

    def cells(self):
        """Iterate ((qx, qy, qz), Material) over every stored cell, boxes expanded."""
        yield from self.data3.items()
        for b, mat in self.data6.items():
            for x in range(b[0], b[3] + 1):
                for y in range(b[1], b[4] + 1):
                    for z in range(b[2], b[5] + 1):
                        yield (x, y, z), mat

    def get_voxels(self):
        """{world position: Material}, every cell expanded to its resolution^3 block (reference data.py:119-133)."""
        r = self.resolution
        out = {}
        for q, mat in self.cells():
            for x in range(q[0] * r, q[0] * r + r):
                for y in range(q[1] * r, q[1] * r + r):
                    for z in range(q[2] * r, q[2] * r + r):
                        out[(x, y, z)] = mat
        return out
