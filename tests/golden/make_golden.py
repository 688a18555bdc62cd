#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *real* reference.

Runs ONLY in the build container (it needs /root/reference); the GPU box and the
test-suite consume the committed ``*.npz`` / ``*.json`` outputs, never this script's
imports.  Nothing from the reference is copied: the reference modules are imported
in place, driven through their public entry points (``Camera.tile`` /
``Camera.trace``, reference ``init.py:37-150``) and observed through wrappers.

Recipe:
  * a tiny stand-in for the absent ``pygame`` package is registered in ``sys.modules``
    (only ``Surface/set_at``, ``image.tobytes`` and ``time.get_ticks`` are touched by
    the hot path: reference ``init.py:127,146,149`` and ``data.py:306``);
  * ``init.py`` is executed up to (not including) its ``Window()`` construction line
    (reference ``init.py:473``) into a module object, so ``Camera`` is importable
    without opening a display;
  * physics frozen, culling off, ``Window.chunk_update`` (reference
    ``init.py:389-452``) is called unbound on a plain ``store`` to fill ``cam.chunks``.

Outputs (all little-endian, numpy ``.npz`` compressed):
  scene_default.npz      flattened default scene (lod0 grid + per-chunk resolution)
  scene_synth64.npz      64^3 hashed synthetic volume (same generator as config 5)
  render_<name>.npz      per-ray end states + per-pixel outputs + event counters
  kat_rng.json           MT19937 known answers from CPython's ``random``
  kat_math.json          glibc sin/cos/pow/radians known answers via CPython ``math``
  ref_timing.json        wall time of the genuine reference (mp.Pool) at config 1

Usage:  python tests/golden/make_golden.py [--only NAME ...]
"""
import argparse
import json
import math
import os
import random
import struct
import sys
import time
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# pygame stand-in (own code; the real package is not installable offline)
# ----------------------------------------------------------------------------
class _Surface:
    def __init__(self, size, flags=0):
        self.size = tuple(size)
        self.px = {}

    def set_at(self, xy, color):
        self.px[tuple(xy)] = tuple(color)


def _install_pygame_stub():
    pg = types.ModuleType("pygame")
    pg.SRCALPHA = 0x00010000
    pg.Surface = _Surface
    pg.image = types.SimpleNamespace(tobytes=lambda surf, fmt: surf)  # hand the surface back
    pg.time = types.SimpleNamespace(get_ticks=lambda: 0)
    sys.modules["pygame"] = pg
    return pg


# ----------------------------------------------------------------------------
# Load the reference in place
# ----------------------------------------------------------------------------
def load_reference():
    _install_pygame_stub()
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.argv = ["init.py", "default"]
    import data  # noqa: E402  (reference module)
    import lib  # noqa: E402
    src = open(os.path.join(REF, "init.py")).read()
    src = src[: src.rindex("# Create the main window")]
    mod = types.ModuleType("refinit")
    mod.__file__ = os.path.join(REF, "init.py")
    sys.modules["refinit"] = mod
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    return data, lib, mod


def set_config(data, **kw):
    """Mutate data.settings and recompute the derived fields (reference data.py:64-77)."""
    s = data.settings
    for k, v in kw.items():
        setattr(s, k, v)
    s.window = s.width, s.height
    s.proportions = ((s.width + s.height) / 2) / max(s.width, s.height)
    s.chunk_radius = round(s.chunk_size / 2)
    s.pixels = [[] for _ in range(s.threads)]
    for x in range(s.width):
        for y in range(s.height):
            s.pixels[(x ^ y) % s.threads].append((x, y))


RENDER_KEYS = ["width", "height", "samples", "static", "shutter", "falloff", "chunk_size", "chunk_lod",
               "fov", "dof", "dist_min", "dist_max", "max_light", "max_bounces", "lod_bounces",
               "lod_samples", "lod_random", "lod_edge", "threads", "proportions", "chunk_radius"]


def settings_dict(data):
    return {k: getattr(data.settings, k) for k in RENDER_KEYS}


# ----------------------------------------------------------------------------
# Event counters through wrappers (reference code is not modified)
# ----------------------------------------------------------------------------
class Counters:
    FIELDS = ["lookup", "nbr", "resnap", "chunk_get", "hit", "draw", "adv", "broke"]

    def __init__(self):
        self.reset()

    def reset(self):
        for f in self.FIELDS:
            setattr(self, f, 0)
        self.pending_hit = False

    def vector(self):
        return [getattr(self, f) for f in self.FIELDS]


def instrument(data, lib, mod, cnt, rays):
    """Wrap the reference's callables so that per-ray event counts can be read."""
    frame_get = data.Frame.get_voxel
    snapped = lib.vec3.snapped
    vmul = lib.vec3.__mul__
    rnd = random.random
    trace = mod.Camera.trace
    gf = sys._getframe

    def get_voxel_w(self, pos):
        ln = gf(1).f_lineno
        if ln == 77:
            cnt.lookup += 1
        elif ln in (103, 104, 105):
            cnt.nbr += 1
        return frame_get(self, pos)

    def snapped_w(self, unit):
        ln = gf(1).f_lineno
        name = gf(1).f_code.co_name
        if name == "trace" and ln == 68:
            cnt.resnap += 1
        elif name == "chunk_get":
            cnt.chunk_get += 1
        return snapped(self, unit)

    def mul_w(self, other):
        f = gf(1)
        if f.f_code.co_name == "trace" and f.f_lineno == 116:
            cnt.adv += 1
            cnt.pending_hit = False
        return vmul(self, other)

    def random_w():
        cnt.draw += 1
        return rnd()

    def trace_w(self, dir_x, dir_y, detail):
        keep_draw = cnt.draw  # the lod_random draw happened in tile() just before
        cnt.reset()
        cnt.draw = keep_draw
        ray = trace(self, dir_x, dir_y, detail)
        cnt.broke = 1 if cnt.pending_hit else 0
        rays.append((dir_x, dir_y, detail, ray, cnt.vector()))
        cnt.draw = 0
        return ray

    data.Frame.get_voxel = get_voxel_w
    lib.vec3.snapped = snapped_w
    lib.vec3.__mul__ = mul_w
    random.random = random_w
    mod.Camera.trace = trace_w

    def wrap_material(m):
        f = m.function

        def fn(ray, mat, settings, _f=f):
            cnt.hit += 1
            cnt.pending_hit = True
            return _f(ray, mat, settings)
        m.function = fn
        m._orig_function = f
    return wrap_material


# ----------------------------------------------------------------------------
# Scene flattening (from the reference's own cam.chunks Frames)
# ----------------------------------------------------------------------------
def frame_points(fr):
    """Yield ((qx,qy,qz), material) for every stored cell of a reference Frame."""
    for q, m in fr.data3.items():
        yield q, m
    for b, m in fr.data6.items():
        for x in range(b[0], b[3] + 1):
            for y in range(b[1], b[4] + 1):
                for z in range(b[2], b[5] + 1):
                    yield (x, y, z), m


def flatten_chunks(chunks, chunk_size, mat_ids):
    """chunks: {(x,y,z)->Frame}.  Returns origin, dims(chunks), present, res, grid[u8, X][Y][Z]."""
    keys = np.array(sorted(chunks.keys()), dtype=np.int64)
    lo = keys.min(0)
    hi = keys.max(0) + chunk_size
    dims = (hi - lo) // chunk_size
    grid = np.zeros(tuple(dims * chunk_size), np.uint8)
    present = np.zeros(tuple(dims), np.uint8)
    res = np.zeros(tuple(dims), np.uint8)
    for post, fr in chunks.items():
        c = (np.array(post) - lo) // chunk_size
        present[tuple(c)] = 1
        res[tuple(c)] = fr.resolution
        for q, m in frame_points(fr):
            p = np.array(q) * fr.resolution
            assert np.all(p >= np.array(post)) and np.all(p < np.array(post) + chunk_size), (post, q)
            grid[tuple(p - lo)] = mat_ids[id(m)]
    return lo, dims, present, res, grid


MAT_PROPS = ["r", "g", "b", "roughness", "absorption", "ior", "energy"]


def material_table(mats):
    t = np.zeros((len(mats), 7), np.float64)
    for i, m in enumerate(mats):
        t[i] = [m.albedo.r, m.albedo.g, m.albedo.b, m.roughness, m.absorption, m.ior, m.energy]
    return t


# ----------------------------------------------------------------------------
# Rendering through the reference and recording
# ----------------------------------------------------------------------------
RAY_F = ["x", "y", "s", "detail", "r", "g", "b", "alpha", "energy", "step", "life", "bounces",
         "px", "py", "pz", "vx", "vy", "vz", "ntrav"] + ["c_" + f for f in Counters.FIELDS]


def render(data, mod, cam, rays, name, per_ray=True, threads_lists=True):
    s = data.settings
    W, H = s.width, s.height
    t0 = time.time()
    rays.clear()
    images = []
    trav = []
    for t in range(s.threads):
        surf, traversed, th = cam.tile(t, 0)
        assert th == t
        images.append(surf)
        trav.append(traversed)
    dt = time.time() - t0
    # per-pixel float means exactly as handed to Surface.set_at (reference init.py:145-146)
    pix = np.full((H, W, 4), np.nan, np.float64)
    owner = np.full((H, W), -1, np.int32)
    for t, surf in enumerate(images):
        for (x, y), c in surf.px.items():
            assert owner[y, x] == -1
            owner[y, x] = t
            pix[y, x] = c
    assert (owner >= 0).all()
    # per-ray records in call order
    rec = np.zeros((len(rays), len(RAY_F)), np.float64)
    last = None
    sidx = 0
    for i, (dx, dy, detail, ray, cv) in enumerate(rays):
        x = round((dx + 1) / 2 * W)
        y = round((dy + 1) / 2 * H)
        assert -1 + (x / W) * 2 == dx and -1 + (y / H) * 2 == dy
        sidx = sidx + 1 if last == (x, y) else 0
        last = (x, y)
        alpha = round(min(1, ray.energy + s.shutter) * 255)
        rec[i] = [x, y, sidx, detail, ray.color.r, ray.color.g, ray.color.b, alpha, ray.energy, ray.step,
                  ray.life, ray.bounces, ray.pos.x, ray.pos.y, ray.pos.z, ray.vel.x, ray.vel.y, ray.vel.z,
                  len(ray.traversed)] + cv
    out = dict(
        settings=np.frombuffer(json.dumps(settings_dict(data)).encode(), np.uint8),
        cam_pos=np.array([cam.pos.x, cam.pos.y, cam.pos.z], np.float64),
        cam_rot=np.array([cam.rot.x, cam.rot.y, cam.rot.z, cam.rot.w], np.float64),
        cam_lens=np.array([cam.lens], np.float64),
        pix_mean=pix, owner=owner.astype(np.int8 if s.threads < 128 else np.int32),
        counters_total=rec[:, -len(Counters.FIELDS):].sum(0).astype(np.int64),
        counter_names=np.array(Counters.FIELDS),
        n_rays=np.array([len(rays)], np.int64),
    )
    if threads_lists:
        for t, tr in enumerate(trav):
            out["traversed_t%d" % t] = np.array([[float(v) for v in p] for p in tr], np.float64).reshape(-1, 3)
    if per_ray:
        out["rays"] = rec
        out["ray_fields"] = np.array(RAY_F)
    else:
        # compact per-sample integer results + energy only
        out["ray_rgba"] = rec[:, [0, 1, 2, 4, 5, 6, 7]].astype(np.int32)
        out["ray_energy"] = rec[:, 8].copy()
    path = os.path.join(OUT, "render_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("  %-14s %6d rays  %.1fs  -> %s (%.0f KB)" % (name, len(rays), dt, os.path.basename(path),
                                                       os.path.getsize(path) / 1024), flush=True)


def murmur_fmix32(h):
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def synth_id(x, y, z, n):
    """Synthetic dense-volume generator of BASELINE config 5, edge length n."""
    half = n // 2
    h = murmur_fmix32(((x + half) + n * ((y + half) + n * (z + half))) ^ 0x5EED5EED)
    if (h & 0xFFFF) >= 1311:
        return 0
    return 1 + ((h >> 16) % 13)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--skip-timing", action="store_true")
    args = ap.parse_args()
    want = (lambda n: args.only is None or n in args.only)

    data, lib, mod = load_reference()
    s = data.settings
    print("reference loaded: %d objects" % len(data.objects))

    # ---- KATs that need no scene ------------------------------------------------
    if want("kat"):
        kat = {}
        seeds = [1, 2, 6, 7, 74703609, 4294967295, 4294967296, 4294967297, 66355200, 268435456,
                 (1 << 40) + 12345, (1 << 64) - 1, (1 << 64) + 5, 3 * 5 * 7]
        for sd in seeds:
            random.seed(sd)
            kat[str(sd)] = [random.random().hex() for _ in range(130)]
        json.dump(kat, open(os.path.join(OUT, "kat_rng.json"), "w"))
        rng = random.Random(12345)
        km = {"sin": [], "cos": [], "pow": [], "radians": []}
        for _ in range(4000):
            a = rng.uniform(-1.7, 1.7)
            km["sin"].append([a.hex(), math.sin(a).hex()])
            km["cos"].append([a.hex(), math.cos(a).hex()])
            d = rng.uniform(-200, 200)
            km["radians"].append([d.hex(), math.radians(d).hex()])
        for b in [1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 6.0, 7.5, 9.0, 10.0, 16.0, 81.0]:
            for e in [1.25, 1.0, 1.5, 2.0, 1.1, 1.75, 3.0]:
                km["pow"].append([float(b).hex(), float(e).hex(), (float(b) ** float(e)).hex()])
        for _ in range(3000):
            b = rng.uniform(1.0, 40.0)
            e = rng.uniform(1.0, 3.0)
            km["pow"].append([b.hex(), e.hex(), (b ** e).hex()])
        json.dump(km, open(os.path.join(OUT, "kat_math.json"), "w"))
        print("KATs written")

    # ---- build the default scene through the reference's own chunk builder ------
    cam = mod.Camera()
    cam.pos = data.player.cam_pos
    cam.rot = data.player.cam_rot
    for obj in data.objects.values():
        obj.physics = False
        obj.update(cam.pos)
    s.culling = False
    set_config(data, threads=1)
    win = lib.store(timer=0, traversed=[[]] * s.threads, chunks={}, chunks_objects={}, cam=cam)
    t0 = time.time()
    mod.Window.chunk_update(win, 1.0)
    print("chunk_update: %.1fs, %d camera chunks" % (time.time() - t0, len(cam.chunks)))

    # material ids 1..13 follow the definition order in the mod script (0 = empty)
    modinit = sys.modules["mods.default.init"]
    mats, mat_ids, names = [], {}, []
    for k, v in vars(modinit).items():
        if isinstance(v, data.Material) and id(v) not in mat_ids:
            mats.append(v)
            names.append(k[4:] if k.startswith("mat_") else k)
            mat_ids[id(v)] = len(mats)
    for post in win.chunks:
        for lodf in win.chunks[post]:
            for _, m in frame_points(lodf):
                assert id(m) in mat_ids
    cs = s.chunk_size
    lo, dims, present, res, grid_cam = flatten_chunks(cam.chunks, cs, mat_ids)
    lo0, dims0, present0, _, grid0 = flatten_chunks({p: f[0] for p, f in win.chunks.items()}, cs, mat_ids)
    assert (lo0 == lo).all() and (dims0 == dims).all() and (present0 == present).all()
    # the camera grid must be the lod0 grid sub-sampled at multiples of the chunk resolution
    derived = np.zeros_like(grid0)
    for cx in range(dims[0]):
        for cy in range(dims[1]):
            for cz in range(dims[2]):
                if not present[cx, cy, cz]:
                    continue
                r = int(res[cx, cy, cz])
                sl = np.s_[cx * cs:(cx + 1) * cs, cy * cs:(cy + 1) * cs, cz * cs:(cz + 1) * cs]
                blk = grid0[sl]
                ax = [(np.arange(cs) + int(lo[a]) + [cx, cy, cz][a] * cs) % r == 0 for a in range(3)]
                mask = ax[0][:, None, None] & ax[1][None, :, None] & ax[2][None, None, :]
                derived[sl] = np.where(mask, blk, 0)
    assert (derived == grid_cam).all(), "camera LOD frames are not a sub-sampling of lod0"
    if want("scene"):
        np.savez_compressed(
            os.path.join(OUT, "scene_default.npz"),
            origin=lo.astype(np.int64), dims=dims.astype(np.int64), chunk_size=np.array([cs], np.int64),
            present=present, res=res, grid_lod0=grid0,
            materials=material_table(mats), material_names=np.array(names),
            material_props=np.array(MAT_PROPS),
            cam_pos=np.array([cam.pos.x, cam.pos.y, cam.pos.z], np.float64),
            cam_rot=np.array([cam.rot.x, cam.rot.y, cam.rot.z, cam.rot.w], np.float64),
            cam_lens=np.array([cam.lens], np.float64),
            settings=np.frombuffer(json.dumps(settings_dict(data)).encode(), np.uint8),
        )
        print("scene_default: %d chunks, %d voxels (lod0), %d materials: %s" % (
            present.sum(), (grid0 > 0).sum(), len(mats), names))

    # ---- timing of the genuine reference (its own mp.Pool dispatch shape) -------
    if want("timing") and not args.skip_timing:
        import multiprocessing as mp
        res_t = {}
        for T in (1, 8):
            set_config(data, width=96, height=54, samples=1, max_bounces=2, threads=T)
            pool = mp.Pool(T)
            best = 1e9
            for rep in range(3):
                t0 = time.time()
                hs = [pool.apply_async(cam.tile, (t, 0)) for t in range(T)]
                outs = [h.get() for h in hs]
                best = min(best, time.time() - t0)
            pool.close()
            pool.join()
            res_t["pool%d_s_per_frame" % T] = best
            res_t["pool%d_primary_rays_per_s" % T] = 96 * 54 / best
        res_t["cpu"] = [l for l in open("/proc/cpuinfo") if "model name" in l][0].split(":")[1].strip()
        res_t["cores"] = os.cpu_count()
        res_t["config"] = "mods/default 96x54 spp1 max_bounces2, culling off, physics frozen"
        json.dump(res_t, open(os.path.join(OUT, "ref_timing.json"), "w"), indent=1)
        print("reference timing:", res_t)

    # ---- instrumented renders ----------------------------------------------------
    cnt = Counters()
    rays = []
    wrap_material = instrument(data, lib, mod, cnt, rays)
    for m in mats:
        wrap_material(m)

    base = dict(width=64, height=48, samples=1, max_bounces=2, threads=1, dof=0.5, lod_edge=0.25,
                lod_random=0.25, lod_samples=0.5, lod_bounces=0.5, dist_max=192, dist_min=0, max_light=1,
                falloff=0.25, shutter=0.25, static=True)

    def run(name, per_ray=True, cam_=None, **kw):
        if not want(name):
            return
        cfg = dict(base)
        cfg.update(kw)
        set_config(data, **cfg)
        render(data, mod, cam_ or cam, rays, name, per_ray=per_ray)

    print("renders:")
    run("g64")                                                        # config.cfg defaults exactly
    run("c1", width=96, height=54)                                    # BASELINE config 1
    run("c1_t8", width=96, height=54, threads=8, per_ray=False)       # config 1 with the (x^y)%8 partition
    run("c1_mb4", width=96, height=54, max_bounces=4, per_ray=False)  # config 2 shape, small
    run("c3small", width=48, height=27, samples=8, max_bounces=8)     # config 3 shape, small
    run("c3_96", width=96, height=54, samples=8, max_bounces=8, per_ray=False)
    run("nolod", samples=2, dof=0, lod_edge=0, lod_random=0, lod_samples=0, lod_bounces=0, max_bounces=8)
    run("dmin", dist_min=3, samples=3, max_bounces=3, max_light=0.5, falloff=0.5, shutter=0.1)
    # rotated, non-integer camera inside the same chunk set
    cam2 = mod.Camera()
    cam2.chunks = cam.chunks
    cam2.pos = lib.vec3(-10.3, 5.7, 3.2)
    cam2.rot = lib.vec3(10, 30, -20).quaternion()
    run("rot", cam_=cam2, samples=2, max_bounces=4)
    # camera far outside the scene box: exercises void skipping and traversed chunks outside the grid
    cam3 = mod.Camera()
    cam3.chunks = cam.chunks
    cam3.pos = lib.vec3(-100.5, 40.25, -90.75)
    cam3.rot = lib.vec3(0, 40, -15).quaternion()
    run("outside", cam_=cam3, samples=1, max_bounces=4)
    # the ray that starts exactly on the origin keeps chunk=None until it leaves [0,0]^3 (init.py:46,67)
    cam4 = mod.Camera()
    cam4.chunks = cam.chunks
    cam4.pos = lib.vec3(0, 0, 0)
    cam4.rot = lib.vec3(0, 180, 0).quaternion()
    run("origin", cam_=cam4, width=32, height=24, samples=1, max_bounces=4)

    # ---- culling feedback loop: chunk selection + LOD over consecutive frames (init.py:389-393, 447-452) ------
    if want("culling"):
        cfg = dict(base)
        cfg.update(width=40, height=30, samples=1, max_bounces=2)
        set_config(data, **cfg)
        s.culling = True
        camc = mod.Camera()
        camc.pos = lib.vec3(cam.pos.x, cam.pos.y, cam.pos.z)
        camc.rot = cam.rot
        winc = lib.store(timer=0, traversed=[[]], chunks=win.chunks, chunks_objects=win.chunks_objects, cam=camc)
        outc = dict(settings=np.frombuffer(json.dumps(settings_dict(data)).encode(), np.uint8),
                    cam_pos=np.array([camc.pos.x, camc.pos.y, camc.pos.z], np.float64),
                    cam_rot=np.array([camc.rot.x, camc.rot.y, camc.rot.z, camc.rot.w], np.float64),
                    cam_lens=np.array([camc.lens], np.float64), chunk_lod=np.array([s.chunk_lod], np.int64),
                    dist_max=np.array([s.dist_max], np.float64))
        moves = [(0, 0, 0), (0, 0, 0), (0, 0, 0), (5.5, -2.25, 30.0), (0, 0, 0)]
        for it, mv in enumerate(moves):
            camc.pos = lib.vec3(camc.pos.x + mv[0], camc.pos.y + mv[1], camc.pos.z + mv[2])
            mod.Window.chunk_update(winc, 1.0)              # selection with the previous frame's traversed list
            pres = np.zeros(tuple(dims), np.uint8)
            rs = np.zeros(tuple(dims), np.uint8)
            for post, fr in camc.chunks.items():
                c = (np.array(post) - lo) // cs
                pres[tuple(c)] = 1
                rs[tuple(c)] = fr.resolution
            rays.clear()
            surf, traversed, _ = camc.tile(0, 0)
            pix = np.zeros((s.height, s.width, 4))
            for (x, y), c in surf.px.items():
                pix[y, x] = c
            winc.traversed = [traversed]
            outc["pos_%d" % it] = np.array([camc.pos.x, camc.pos.y, camc.pos.z], np.float64)
            outc["present_%d" % it] = pres
            outc["res_%d" % it] = rs
            outc["pix_%d" % it] = pix
            outc["traversed_%d" % it] = np.array([[float(v) for v in p] for p in traversed], np.float64).reshape(-1, 3)
            print("  culling frame %d: %d camera chunks, %d traversed" % (it, pres.sum(), len(traversed)))
        np.savez_compressed(os.path.join(OUT, "culling_sequence.npz"), **outc)
        s.culling = False

    # ---- world build: Goxel text import -> Sprite -> Object -> world chunks (data.py:253-427, 430-494, 589-600;
    #      init.py:398-444) on synthetic assets (tests/golden/assets, own data) ------------------------------------
    if want("world"):
        assets = os.path.join(OUT, "assets")
        saved_objects = dict(data.objects)
        data.objects.clear()
        wm = []
        for i, (alb, rough, absb, ior, en) in enumerate([((255, 0, 0), 0.5, 1.0, 1.0, 0.0), ((0, 255, 0), 0.0, 0.5, 0.5, 0.0),
                                                         ((0, 0, 255), 0.25, 1.5, 1.0, 2.0), ((127, 127, 127), 0.1, 0.25, 0.0, 0.0)]):
            wm.append(data.Material(function=lib.material, albedo=lib.rgb(*alb), roughness=rough, absorption=absb, ior=ior,
                                    energy=en, solidity=1, weight=0.001, friction=0.1, elasticity=0.5))
        cmap = {"ff0000": wm[0], "00ff00": wm[1], "0000ff": wm[2], "7f7f7f": wm[3]}
        specs = [  # file, sprite size as given, lod, position, rotation
            ("slab.txt", (10, 6, 8), 0, (0, 0, 0), (0, 0, 0)),
            ("cube.txt.gz", (12, 12, 12), 0, (9, 2, -3), (0, 90, 0)),       # overlaps the slab, rotated about Y
            ("cube.txt.gz", (12, 12, 12), 0, (-20, 5, 14), (90, 0, 180)),
            ("odd.txt", (7, 5, 9), 0, (3.5, -9.25, 20.0), (0, 180, 0)),     # odd size -> rounded up; non-integer position
            ("slab.txt", (10, 6, 8), 1, (-30, -10, -30), (0, 0, 0)),        # sprite lod 1: frame resolution 2
            ("cube.txt.gz", (12, 12, 12), 0, (400, 0, 0), (0, 0, 0)),       # beyond dist_max + size: not visible
        ]
        objs = []
        for fn, size, lod, pos, rot in specs:
            spr = data.Sprite(size=lib.vec3(*size), frames=1, lod=lod)
            spr.load([os.path.join(assets, fn)], cmap)
            ob = data.Object(pos=lib.vec3(*pos), rot=lib.vec3(*rot), vel=lib.vec3(0, 0, 0), physics=False)
            ob.set_sprite(spr)
            objs.append(ob)
        camw = mod.Camera()
        camw.pos = lib.vec3(2.0, 3.0, -40.0)
        camw.rot = lib.quaternion(0, 0, 0, 1)
        for ob in objs:
            ob.update(camw.pos)
        cfg = dict(base)
        cfg.update(width=48, height=36, samples=2, max_bounces=4)
        set_config(data, **cfg)
        s.culling = False
        winw = lib.store(timer=0, traversed=[[]], chunks={}, chunks_objects={}, cam=camw)
        mod.Window.chunk_update(winw, 1.0)
        wids = {id(m): i + 1 for i, m in enumerate(wm)}
        wlo, wdims, wpres, _, wgrid = flatten_chunks({p: f[0] for p, f in winw.chunks.items()}, cs, wids)
        _, _, cpres, cres, _ = flatten_chunks(camw.chunks, cs, wids)
        for m in wm:
            wrap_material(m)
        rays.clear()
        surf, travw, _ = camw.tile(0, 0)
        pixw = np.zeros((s.height, s.width, 4))
        for (x, y), c in surf.px.items():
            pixw[y, x] = c
        np.savez_compressed(
            os.path.join(OUT, "world_build.npz"),
            spec_files=np.array([sp[0] for sp in specs]), spec_size=np.array([sp[1] for sp in specs], np.float64),
            spec_lod=np.array([sp[2] for sp in specs], np.int64), spec_pos=np.array([sp[3] for sp in specs], np.float64),
            spec_rot=np.array([sp[4] for sp in specs], np.float64),
            colours=np.array(list(cmap.keys())), materials=material_table(wm),
            sprite_size=np.array([[o.sprite.size.x, o.sprite.size.y, o.sprite.size.z] for o in objs], np.int64),
            obj_mins=np.array([[o.mins.x, o.mins.y, o.mins.z] for o in objs], np.float64),
            obj_maxs=np.array([[o.maxs.x, o.maxs.y, o.maxs.z] for o in objs], np.float64),
            obj_visible=np.array([bool(o.visible) for o in objs]),
            origin=wlo.astype(np.int64), dims=wdims.astype(np.int64), present=wpres, grid_lod0=wgrid,
            cam_present=cpres, cam_res=cres,
            cam_pos=np.array([camw.pos.x, camw.pos.y, camw.pos.z]), cam_rot=np.array([0.0, 0, 0, 1]),
            cam_lens=np.array([camw.lens]), pix=pixw,
            traversed=np.array([[float(v) for v in p] for p in travw], np.float64).reshape(-1, 3),
            settings=np.frombuffer(json.dumps(settings_dict(data)).encode(), np.uint8))
        print("world_build: %d objects (%d visible), %d world chunks, %d voxels" % (
            len(objs), sum(bool(o.visible) for o in objs), int(wpres.sum()), int((wgrid > 0).sum())))

        # ---- re-draws (init.py:398-429): which object wins an overlap after objects redraw separately and then together.
        #      The slab (objs[0]) and the first cube (objs[1]) overlap; data.objects holds them in that order.
        seq = {}

        def snap(tag):
            lo, dims, pres, _, grid = flatten_chunks({p: f[0] for p, f in winw.chunks.items() if f}, cs, wids)
            seq["origin_" + tag], seq["dims_" + tag], seq["grid_" + tag] = lo.astype(np.int64), dims.astype(np.int64), grid
            seq["order_" + tag] = np.array([[o.id for o in objs].index(k) for k in winw.chunks_objects.keys()], np.int64)

        snap("0")
        objs[0].redraw = True                       # tick 1: the slab redraws alone -> it goes behind the cube in the dict
        mod.Window.chunk_update(winw, 1.0)
        snap("1")
        objs[0].redraw = True                       # tick 2: both redraw (the cube moved by one voxel)
        objs[1].move(objs[1].pos + lib.vec3(1, 0, 0))
        objs[1].update(camw.pos)
        mod.Window.chunk_update(winw, 1.0)
        snap("2")
        seq["cube_pos_2"] = np.array([objs[1].pos.x, objs[1].pos.y, objs[1].pos.z], np.float64)
        np.savez_compressed(os.path.join(OUT, "world_update.npz"), **seq)
        print("world_update: merge orders", [seq["order_%d" % k].tolist() for k in range(3)])
        data.objects.clear()
        data.objects.update(saved_objects)

    # ---- synthetic 64^3 hashed volume (config 5 generator, small) -----------------
    if want("synth64"):
        n = 64
        half = n // 2
        synth_mats = mats[:13]
        assert len(synth_mats) == 13
        chunks = {}
        grid = np.zeros((n, n, n), np.uint8)
        for cx in range(-half, half, cs):
            for cy in range(-half, half, cs):
                for cz in range(-half, half, cs):
                    fr = data.Frame(packed=False, resolution=1)
                    for x in range(cx, cx + cs):
                        for y in range(cy, cy + cs):
                            for z in range(cz, cz + cs):
                                i = synth_id(x, y, z, n)
                                if i:
                                    fr.data3[(x, y, z)] = synth_mats[i - 1]
                                    grid[x + half, y + half, z + half] = i
                    chunks[(cx, cy, cz)] = fr
        cam5 = mod.Camera()
        cam5.chunks = chunks
        cam5.pos = lib.vec3(0.5, 0.5, 0.5)
        cam5.rot = lib.quaternion(0, 0, 0, 1)
        np.savez_compressed(
            os.path.join(OUT, "scene_synth64.npz"),
            origin=np.array([-half] * 3, np.int64), dims=np.array([n // cs] * 3, np.int64),
            chunk_size=np.array([cs], np.int64), present=np.ones((n // cs,) * 3, np.uint8),
            res=np.ones((n // cs,) * 3, np.uint8), grid_lod0=grid,
            materials=material_table(synth_mats), material_names=np.array(names[:13]),
            material_props=np.array(MAT_PROPS),
            cam_pos=np.array([0.5, 0.5, 0.5]), cam_rot=np.array([0.0, 0, 0, 1]), cam_lens=np.array([cam5.lens]),
        )
        cfg = dict(base)
        cfg.update(width=64, height=64, samples=2, max_bounces=8, dist_max=64, dof=0, lod_edge=0, lod_random=0,
                   lod_samples=0, lod_bounces=0)
        set_config(data, **cfg)
        render(data, mod, cam5, rays, "synth64")


if __name__ == "__main__":
    main()
