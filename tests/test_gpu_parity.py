"""GPU parity tests: the HIP path (through the C ABI, via python_raytracer_amd.Camera) against
  (1) the CPU oracle in its portable-libm mode        -> every field of every ray BIT-EXACT,
  (2) the golden vectors produced by the real reference -> integer fields bit-exact, binary64 fields within
      1e-11 absolute (glibc's sin/cos/pow are not correctly rounded in ~0.1% of calls),
  (3) size-independent properties at BASELINE's full sizes.
Tolerance stated by the north star: fp32 RGB within 1e-5, integer pixel / hit indices bit-exact.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from gpu_util import camera_for, settings_store

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["pool", "lanes", "pool-ahead", "lanes-ahead"])
def frame_march(request, monkeypatch):
    """Every test runs with both frame kernels: march_pool_kernel (rays regrouped between lanes through LDS; the library
    only picks it for launches of 5 Mi rays and more) and march_kernel (one ray per lane) -- and each of them with the
    march step that stops its look-ahead at chunk borders (shipped) and the one that looks ahead across them
    (march_step_w, VRT_WADDR=1: a measured variant for scenes whose blocks lie in table order).  VRT_POOL,
    VRT_POOL_MIN_RAYS and VRT_WADDR are read at every launch.  Tests marked `one_march` (they start their own processes
    or do not render frames) run once."""
    if request.node.get_closest_marker("one_march") and request.param != "pool":
        pytest.skip("runs once")
    monkeypatch.setenv("VRT_POOL", "1" if request.param.startswith("pool") else "0")
    monkeypatch.setenv("VRT_POOL_MIN_RAYS", "0")
    monkeypatch.setenv("VRT_WADDR", "1" if request.param.endswith("-ahead") else "0")
    return request.param

GOLD = ["g64", "c1", "c3small", "nolod", "dmin", "rot", "outside", "origin", "synth64"]


def scene_for(name):
    return ol.synth64_scene() if name == "synth64" else ol.default_scene()


def settings_of(g):
    s = g["settings"]
    return ol.make_settings(**{k: s[k] for k in ol.DEFAULT_SETTINGS if k in s})


def gpu_render(name, **kw):
    g = ol.load_render(name)
    st = settings_of(g)
    sc = scene_for(name)
    cam = camera_for(sc, settings_store(st), g["cam_pos"], g["cam_rot"], g["cam_lens"][0])
    r = cam.render(0, want_rays=True, want_ray_rgba=True, **kw)
    return g, st, sc, cam, r


def check_frame_march(cam, o, cs, which, lookahead=None, **kw):
    """The same frame WITHOUT ray records: `want_rays` selects the recording march_kernel whatever VRT_POOL says, so this is
    the render that runs the frame kernel the fixture names -- march_pool_kernel under "pool" (asserted: its workgroups
    count themselves in stats[12]), march_kernel under "lanes".  Per-sample colours, fp32 means, event counters and the
    traversed list against the oracle."""
    r = cam.render(0, want_ray_rgba=True, **kw)
    groups = int(r.stats[12]) & 0xffffffff     # (bits 32+: the workgroups that took their rays as tiles)
    assert (groups > 0) if which.startswith("pool") else (groups == 0), (which, groups)
    if lookahead is not None:   # did the march step look ahead across chunk borders (march_step_w)?
        assert (int(r.stats[14]) > 0) == bool(lookahead), (lookahead, int(r.stats[14]))
    assert np.array_equal(r.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32))
    assert (r.stats[:8] == o["counters"]).all(), (r.stats[:8], o["counters"])
    assert np.array_equal(np.array(r.traversed(cs), np.int64).reshape(-1, 3), np.asarray(o["traversed"]).reshape(-1, 3))
    rays = o["rays"]
    where = {(int(x), int(y)): i for i, (x, y) in enumerate(r.pixels)}
    slot = np.array([where[(int(x), int(y))] for x, y in zip(rays["x"], rays["y"])], np.int64) * r.max_samples + rays["s"]
    packed = (rays["color"][:, 0].astype(np.uint32) | (rays["color"][:, 1].astype(np.uint32) << 8) |
              (rays["color"][:, 2].astype(np.uint32) << 16) | (rays["alpha"].astype(np.uint32) << 24))
    got = r.ray_rgba.cpu().numpy().view(np.uint32)
    assert np.array_equal(got[slot], packed)
    # ... and once more without the settled-cell bitmap (VRT_TRAV_LDS=0, read at every launch): what traversed boxes too large
    # for one get -- every visit reads its cell's key, and the kernel instances that compare it after the voxel reads run
    # -- and, where the window's geometry allows (the pool kernel, a power-of-two run of pixels per hand-out that divides the
    # height), with the rays handed out as square tiles in Morton order from eight heads (VRT_TILED=2: tile_ticket)
    os.environ["VRT_TRAV_LDS"], os.environ["VRT_DEFER_VISIT"], os.environ["VRT_TILED"] = "0", "2", "2"   # (2: also over scenes that fit the caches)
    try:
        r2 = cam.render(0, want_ray_rgba=True, **kw)
    finally:
        del os.environ["VRT_TRAV_LDS"], os.environ["VRT_DEFER_VISIT"], os.environ["VRT_TILED"]
    assert np.array_equal(r2.ray_rgba.cpu().numpy(), r.ray_rgba.cpu().numpy()) and (r2.stats[:9] == r.stats[:9]).all()
    assert np.array_equal(r2.traversed_keys.cpu().numpy(), r.traversed_keys.cpu().numpy())
    # ... with the settled bitmap of boxes too large for one of their own: over the 32^3 cells around the camera only
    # (VRT_TRAV_WINDOW=2 uses it for every box of at least 32^3 cells; smaller boxes render as before)
    # -- together with the key comparison behind the voxel reads, as scenes beyond the caches run it
    os.environ["VRT_TRAV_WINDOW"], os.environ["VRT_DEFER_VISIT"] = "2", "2"
    try:
        r4 = cam.render(0, want_ray_rgba=True, **kw)
    finally:
        del os.environ["VRT_TRAV_WINDOW"], os.environ["VRT_DEFER_VISIT"]
    assert np.array_equal(r4.ray_rgba.cpu().numpy(), r.ray_rgba.cpu().numpy()) and (r4.stats[:9] == r.stats[:9]).all()
    assert np.array_equal(r4.traversed_keys.cpu().numpy(), r.traversed_keys.cpu().numpy())
    # ... and once without the cached ray table (Camera.cache_draws = False): the frame's draws are seeded anew and the march
    # works out every ray's lens quaternion and life itself instead of reading raygen_tile_kernel's records -- asserted where
    # the library has such a march (stats[15]: not for resolutions > 2, the look-ahead variant or one record per pixel)
    s = cam._settings()
    fused = (not which.endswith("-ahead") and 1 <= int(cam._c_scene(cam._ensure_scene()).max_resolution) <= 2 and
             (float(s.dof) != 0.0 or float(s.lod_random) != 0.0 or float(s.lod_samples) != 0.0))
    cached, cam.cache_draws = cam.cache_draws, False
    try:
        r3 = cam.render(0, want_ray_rgba=True, **kw)
    finally:
        cam.cache_draws = cached
    if fused:
        assert int(r3.stats[15]) > 0, r3.stats
    assert np.array_equal(r3.ray_rgba.cpu().numpy(), r.ray_rgba.cpu().numpy()) and (r3.stats[:9] == r.stats[:9]).all()
    assert np.array_equal(r3.traversed_keys.cpu().numpy(), r.traversed_keys.cpu().numpy())
    return r


def active(r):
    rays = r.rays
    return rays[rays["s"] >= 0]


# ------------------------------------------------------------------------------------------------- RNG
@pytest.mark.one_march
def test_rng_kernel_matches_cpython_kat():
    import torch
    import ctypes as C
    from python_raytracer_amd import _native as nat
    L = nat.lib()
    kat = json.load(open(os.path.join(ol.GOLDEN, "kat_rng.json")))
    seeds = [int(s) for s in kat if int(s) < 2 ** 64]
    rng = np.random.default_rng(5)
    extra = [int(v) for v in rng.integers(0, 2 ** 63, 3000)] + [0, 1, 2 ** 32 - 1, 2 ** 32, 2 ** 64 - 1]
    all_seeds = seeds + extra
    d_seeds = torch.tensor(np.array(all_seeds, np.uint64).view(np.int64), device="cuda")
    for nd in (2, 3, 8, 31, 32, 64, 113):
        out = torch.zeros((len(all_seeds), nd), dtype=torch.float64, device="cuda")
        nat.check(L.vrt_rng_draws(d_seeds.data_ptr(), len(all_seeds), nd, out.data_ptr(), None), "vrt_rng_draws")
        got = out.cpu().numpy().T
        for i, s in enumerate(seeds):
            exp = np.array([float.fromhex(v) for v in kat[str(s)][:nd]])
            assert (got[:, i] == exp).all(), (s, nd)
        for i, s in enumerate(all_seeds):
            if i % 97 == 0 or i >= len(all_seeds) - 5:
                assert (got[:, i] == ol.rng_draws(s, nd)).all(), (s, nd)


@pytest.mark.one_march
def test_rng_full_state_generator_beyond_113_draws():
    """vrt_rng_draws above 113 draws uses the full-state MT19937 (also the third retrace tier): CPython KATs with 700
    draws (tests/golden/kat_rng_long.json) and the oracle for random 64-bit seeds, across the twist at draw 312."""
    import torch
    from python_raytracer_amd import _native as nat
    L = nat.lib()
    kat = json.load(open(os.path.join(ol.GOLDEN, "kat_rng_long.json")))
    seeds = [int(s) for s in kat]
    rng = np.random.default_rng(9)
    all_seeds = seeds + [int(v) for v in rng.integers(0, 2 ** 63, 500)] + [0, 2 ** 32, 2 ** 64 - 1]
    d_seeds = torch.tensor(np.array(all_seeds, np.uint64).view(np.int64), device="cuda")
    for nd in (114, 311, 312, 313, 700, 1024):
        out = torch.zeros((len(all_seeds), nd), dtype=torch.float64, device="cuda")
        nat.check(L.vrt_rng_draws(d_seeds.data_ptr(), len(all_seeds), nd, out.data_ptr(), None), "vrt_rng_draws")
        got = out.cpu().numpy()
        for i, s in enumerate(seeds):
            exp = np.array([float.fromhex(v) for v in kat[str(s)][:nd]])
            assert (got[i, :len(exp)] == exp).all(), (s, nd)
        for i, s in enumerate(all_seeds):
            if i % 31 == 0 or i >= len(all_seeds) - 3:
                assert (got[i] == ol.rng_draws(s, nd)).all(), (s, nd)
    assert L.vrt_rng_draws(d_seeds.data_ptr(), 1, 4097, out.data_ptr(), None) != 0


def test_third_retrace_tier_many_rough_hits(frame_march):
    """Weakly absorbing rough materials with a large bounce budget: rays take more than 37 rough hits, i.e. more
    than the 113 draws of the second tier; the third tier (1024 draws, full-state MT19937) completes them.  Every
    ray bit-exact against the oracle."""
    rng = np.random.default_rng(77)
    cs = 16
    dims = np.array([2, 2, 2])
    origin = np.array([-16, -16, -16], np.int64)
    present = np.ones(tuple(dims), np.uint8)
    res = np.ones(tuple(dims), np.uint8)
    mats = np.array([[200, 180, 160, 1.0, 0.05, 1.0, 0.0], [90, 120, 250, 0.5, 0.05, 0.5, 0.0]])
    grid = np.where(rng.random(tuple(dims * cs)) < 0.35, rng.integers(1, 3, tuple(dims * cs)), 0).astype(np.uint8)
    sc = ol.Scene(origin, dims, cs, present, res, grid, mats)
    st = ol.make_settings(width=40, height=30, samples=2, max_bounces=16.0, chunk_size=cs, dist_max=200, falloff=0.0,
                          max_light=100.0, lod_bounces=0.0)
    pos, q, lens = np.array([0.5, 0.5, 0.5]), np.array([0.0, 0.0, 0.0, 1.0]), 90 * np.pi / 8
    cam = camera_for(sc, settings_store(st), pos, q, lens)
    r = cam.render(0, want_rays=True)
    o = ol.render(sc, st, pos, q, lens, r.pixels, libm=ol.LIBM_PORTABLE)
    got, exp = active(r), o["rays"]
    assert exp["counters"][:, ol.COUNTERS.index("draw")].max() > 113       # the scene really needs the third tier
    assert r.stats[10] == 0
    for f in ("color", "alpha", "counters", "ntrav", "energy", "step", "life", "bounces", "pos", "vel"):
        assert np.array_equal(got[f], exp[f]), f
    assert np.array_equal(r.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32))
    assert (r.stats[:8] == o["counters"]).all()
    check_frame_march(cam, o, cs, frame_march)


# ------------------------------------------------------------------------------------------------- per ray
@pytest.mark.parametrize("name", GOLD)
def test_rays_bit_exact_vs_oracle(name, frame_march):
    g, st, sc, cam, r = gpu_render(name)
    o = ol.render(sc, st, g["cam_pos"], g["cam_rot"], g["cam_lens"][0], r.pixels, libm=ol.LIBM_PORTABLE)
    got, exp = active(r), o["rays"]
    assert len(got) == len(exp) == int(r.stats[8])
    for f in ("x", "y", "s", "color", "alpha", "counters", "ntrav", "detail", "energy", "step", "life", "bounces",
              "pos", "vel"):
        assert np.array_equal(got[f], exp[f]), (f, np.flatnonzero((got[f] != exp[f]).reshape(len(got), -1).any(1))[:5])
    assert (r.stats[:8] == o["counters"]).all()
    assert r.stats[10] == 0 and r.stats[11] == 0
    # image: fp32 means and RGBA8
    assert np.array_equal(r.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32))
    img = r.image_u8.cpu().numpy()
    assert np.array_equal(img[r.pixels[:, 1], r.pixels[:, 0]], o["pix_rgba8"])
    # traversed chunk list in the reference's order
    trav = np.array(r.traversed(st["chunk_size"]), np.int64).reshape(-1, 3)
    assert np.array_equal(trav, o["traversed"])
    # (the fixture scenes are small boxes at resolutions <= 2: laid out in table order, marched across chunk borders)
    check_frame_march(cam, o, st["chunk_size"], frame_march, lookahead=frame_march.endswith("-ahead"))


@pytest.mark.parametrize("name", GOLD)
def test_rays_vs_reference_golden(name):
    """Directly against what the real reference produced (glibc libm): integers exact, fp64 within 1e-11."""
    g, st, sc, cam, r = gpu_render(name)
    F = {k: i for i, k in enumerate(g["ray_fields"])}
    R = g["rays"]
    got = active(r)
    assert len(got) == len(R)
    for f in ("x", "y", "s", "alpha", "ntrav"):
        assert np.array_equal(got[f], R[:, F[f]]), f
    for i, c in enumerate("rgb"):
        assert np.array_equal(got["color"][:, i], R[:, F[c]]), c
    for i, cn in enumerate(ol.COUNTERS):
        assert np.array_equal(got["counters"][:, i], R[:, F["c_" + cn]]), cn
    for f in ("detail", "energy", "step", "life", "bounces"):
        assert np.allclose(got[f], R[:, F[f]], rtol=1e-13, atol=0), f
    for i, c in enumerate("xyz"):
        assert np.allclose(got["pos"][:, i], R[:, F["p" + c]], rtol=0, atol=1e-11)
        assert np.allclose(got["vel"][:, i], R[:, F["v" + c]], rtol=0, atol=1e-13)
    H, W = g["pix_mean"].shape[:2]
    f32 = r.rgba_f32.cpu().numpy()
    ref = g["pix_mean"][r.pixels[:, 1], r.pixels[:, 0]]
    assert np.abs(f32.astype(np.float64) - ref).max() <= 1e-5          # the north star's fp32 tolerance
    assert np.array_equal(f32, ref.astype(np.float32))                 # and in fact identical
    trav = np.array(r.traversed(st["chunk_size"]), np.float64).reshape(-1, 3)
    assert np.array_equal(trav, g["traversed_t0"])
    assert (r.stats[:8] == g["counters_total"]).all()


@pytest.mark.parametrize("name", ["c1_t8", "c1_mb4", "c3_96"])
def test_compact_golden_images(name):
    g = ol.load_render(name)
    st = settings_of(g)
    sc = ol.default_scene()
    T = g["settings"]["threads"]
    st["threads"] = T
    cam = camera_for(sc, settings_store(st), g["cam_pos"], g["cam_rot"], g["cam_lens"][0])
    total = np.zeros(8, np.int64)
    for t in range(T):
        image, traversed, th = cam.tile(t, 0)          # the reference's entry point and return triple
        assert th == t and isinstance(image, bytes) and len(image) == st["width"] * st["height"] * 4
        img = np.frombuffer(image, np.uint8).reshape(st["height"], st["width"], 4)
        own = g["owner"] == t
        assert np.array_equal(img[own], np.trunc(g["pix_mean"][own]).astype(np.uint8))
        assert (img[~own] == 0).all()
        assert np.array_equal(np.array(traversed).reshape(-1, 3), g["traversed_t%d" % t])
        total += cam.last_stats[:8]
    assert (total == g["counters_total"]).all()
    f32 = cam.tile_f32(0).cpu().numpy()
    own = g["owner"] == 0
    assert np.array_equal(f32[own], g["pix_mean"][own].astype(np.float32))


# ------------------------------------------------------------------------------------------------- API surface
@pytest.mark.one_march
def test_trace_single_ray_uses_python_rng_stream():
    import random
    g = ol.load_render("g64")
    st = settings_of(g)
    sc = ol.default_scene()
    cam = camera_for(sc, settings_store(st), g["cam_pos"], g["cam_rot"], g["cam_lens"][0])
    F = {k: i for i, k in enumerate(g["ray_fields"])}
    for idx in (0, 1561, 3071, 490):
        row = g["rays"][idx]
        x, y = int(row[F["x"]]), int(row[F["y"]])
        random.seed((1 + x) * (1 + y))
        first = random.random()                          # tile()'s lod_random draw (reference init.py:139)
        dir_x, dir_y = -1 + (x / st["width"]) * 2, -1 + (y / st["height"]) * 2
        detail = 1 - abs(dir_x * dir_y) * st["lod_edge"]
        ray_detail = detail / 1 * (1 - st["lod_random"] * first)
        assert ray_detail == row[F["detail"]]
        ray = cam.trace(dir_x, dir_y, ray_detail)
        assert ray.color.tuple() == (row[F["r"]], row[F["g"]], row[F["b"]])
        assert ray.step == row[F["step"]] and ray.bounces == row[F["bounces"]]
        assert abs(ray.energy - row[F["energy"]]) < 1e-13 and len(ray.traversed) == row[F["ntrav"]]
        # the global stream advanced by exactly the draws the ray consumed
        random.seed((1 + x) * (1 + y))
        for _ in range(int(row[F["c_draw"]])):
            random.random()
        expect_next = random.random()
        random.seed((1 + x) * (1 + y))
        random.random()
        cam.trace(dir_x, dir_y, ray_detail)
        assert random.random() == expect_next


def test_frame_dict_scene_equals_dense_scene():
    """Camera.chunks of Frame objects (the reference's input form) flattens to the same render as the dense path."""
    from python_raytracer_amd import Camera, Frame, Material
    from python_raytracer_amd.lib import rgb, material, vec3, quaternion
    sc = ol.synth64_scene()
    st = ol.make_settings(width=48, height=40, samples=3, max_bounces=6, dist_max=96)
    mats = [Material(function=material, albedo=rgb(*[int(v) for v in row[:3]]), roughness=row[3], absorption=row[4],
                     ior=row[5], energy=row[6]) for row in sc.materials]
    cam = Camera(settings=settings_store(st))
    res_of = {}
    for cx in range(4):
        for cy in range(4):
            for cz in range(4):
                r = 1 + (cx + 2 * cy + cz) % 3                      # resolutions 1, 2, 3
                post = (cx * 16 - 32, cy * 16 - 32, cz * 16 - 32)
                if (cx, cy, cz) == (3, 3, 3):
                    continue                                       # one missing chunk (void)
                fr = Frame(packed=True, resolution=r)
                blk = sc.grid[cx * 16:cx * 16 + 16, cy * 16:cy * 16 + 16, cz * 16:cz * 16 + 16]
                vox = {}
                for lx, ly, lz in zip(*np.nonzero(blk)):
                    vox[(post[0] + int(lx), post[1] + int(ly), post[2] + int(lz))] = mats[int(blk[lx, ly, lz]) - 1]
                fr.set_voxels(vox, True)
                cam.chunk_set(post, fr)
                res_of[(cx, cy, cz)] = r
    cam.pos, cam.rot = vec3(0.5, 0.5, 0.5), quaternion(0.1, -0.2, 0.05, 0.97)
    r1 = cam.render(0, want_rays=True)
    # oracle on the equivalent dense description
    present = np.ones((4, 4, 4), np.uint8)
    present[3, 3, 3] = 0
    res = np.ones((4, 4, 4), np.uint8)
    for k, v in res_of.items():
        res[k] = v
    used = cam._materials
    remap = np.zeros(14, np.uint8)
    for new, m in enumerate(used):
        remap[1 + mats.index(m)] = new + 1
    table = np.array([[m.albedo.r, m.albedo.g, m.albedo.b, m.roughness, m.absorption, m.ior, m.energy] for m in used])
    grid = ol.Scene.camera_grid(remap[sc.grid], sc.origin, sc.dims, 16, present, res)
    dsc = ol.Scene(sc.origin, sc.dims, 16, present, res, grid, table)
    o = ol.render(dsc, st, [0.5, 0.5, 0.5], [0.1, -0.2, 0.05, 0.97], cam.lens, r1.pixels, libm=ol.LIBM_PORTABLE)
    got = active(r1)
    for f in ("color", "alpha", "counters", "energy", "step", "life", "bounces", "pos", "vel", "ntrav"):
        assert np.array_equal(got[f], o["rays"][f]), f
    assert cam.chunk_get(vec3(-20.5, 3.0, 17.2)) is cam.chunks[(-32, 0, 16)]
    assert cam.chunk_get(vec3(100, 0, 0)) is None
    cam.chunk_set((-32, 0, 16), None)
    assert (-32, 0, 16) not in cam.chunks


def test_edge_cases(frame_march):
    from python_raytracer_amd import Camera
    from python_raytracer_amd.lib import vec3, quaternion
    sc = ol.default_scene()
    # empty pixel list
    st = ol.make_settings(width=16, height=8, threads=32)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    empty = [t for t in range(32) if len(cam.settings.pixels[t]) == 0]
    if empty:
        image, trav, th = cam.tile(empty[0], 0)
        assert image == bytes(16 * 8 * 4) and trav == []
    # no chunks at all: every ray is void-skipped to the sky
    cam2 = Camera(settings=settings_store(ol.make_settings(width=24, height=16, samples=2)))
    cam2.pos, cam2.rot = vec3(3.5, -7.25, 11.0), quaternion(0, 0, 0, 1)
    r = cam2.render(0, want_rays=True)
    esc = ol.Scene([0, 0, 0], [1, 1, 1], 16, np.zeros((1, 1, 1), np.uint8), np.zeros((1, 1, 1), np.uint8),
                   np.zeros((16, 16, 16), np.uint8), np.zeros((0, 7)))
    o = ol.render(esc, ol.make_settings(width=24, height=16, samples=2), [3.5, -7.25, 11.0], [0, 0, 0, 1], cam2.lens,
                  r.pixels, libm=ol.LIBM_PORTABLE)
    got = active(r)
    for f in ("color", "alpha", "counters", "energy", "step", "pos", "vel", "ntrav"):
        assert np.array_equal(got[f], o["rays"][f]), f
    assert r.stats[0] == 0 and r.stats[4] == 0
    check_frame_march(cam2, o, 16, frame_march)
    # background None (reference init.py:119): colour stays un-energised
    from python_raytracer_amd import data
    try:
        data.background = None
        st3 = ol.make_settings(width=32, height=24)
        cam3 = camera_for(sc, settings_store(st3), sc.cam_pos, sc.cam_rot, sc.cam_lens)
        r3 = cam3.render(0, want_rays=True)
        o3 = ol.render(sc, st3, sc.cam_pos, sc.cam_rot, sc.cam_lens, r3.pixels, libm=ol.LIBM_PORTABLE,
                       has_background=False)
        assert np.array_equal(active(r3)["color"], o3["rays"]["color"])
        assert np.array_equal(active(r3)["energy"], o3["rays"]["energy"])
        check_frame_march(cam3, o3, 16, frame_march)
    finally:
        data.background = data.material_background
    # non-static seeding: same nonce -> same image as the oracle with that nonce
    st4 = ol.make_settings(width=32, height=24, samples=2, static=False)
    cam4 = camera_for(sc, settings_store(st4), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    r4 = cam4.render(0, seed_nonce=0x1234567, want_rays=True)
    o4 = ol.render(sc, st4, sc.cam_pos, sc.cam_rot, sc.cam_lens, r4.pixels, libm=ol.LIBM_PORTABLE, seed_nonce=0x1234567)
    assert np.array_equal(active(r4)["color"], o4["rays"]["color"])
    assert np.array_equal(active(r4)["detail"], o4["rays"]["detail"])
    check_frame_march(cam4, o4, 16, frame_march, seed_nonce=0x1234567)
    a, b = cam4.render(0), cam4.render(0)
    assert not np.array_equal(a.rgba_f32.cpu().numpy(), b.rgba_f32.cpu().numpy())   # fresh nonce every call
    # Non-static rays have streams of their own: with static seeds pixels (3, 5) and (5, 3) share (1 + x)(1 + y) and so
    # their draws; in a non-static frame they must not.  The first draw is recovered from the ray's detail
    # (init.py:139: detail / (1 + s * lod_samples) * (1 - lod_random * draw)).
    def first_draws(rays, x, y):
        sel = rays[(rays["x"] == x) & (rays["y"] == y)]
        dx, dy = -1 + (x / st4["width"]) * 2, -1 + (y / st4["height"]) * 2
        det = 1 - abs(dx * dy) * st4["lod_edge"]
        return (1 - sel["detail"] * (1 + sel["s"] * st4["lod_samples"]) / det) / st4["lod_random"]
    st5 = dict(st4, static=True)
    cam5 = camera_for(sc, settings_store(st5), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    r5 = cam5.render(0, want_rays=True)
    assert np.allclose(first_draws(active(r5), 3, 5), first_draws(active(r5), 5, 3), rtol=0, atol=1e-12)
    d35, d53 = first_draws(active(r4), 3, 5), first_draws(active(r4), 5, 3)
    assert len(d35) == len(d53) > 0 and np.abs(d35 - d53).min() > 1e-6


def test_rng_retrace_path(frame_march):
    """Rays that need more than the 32 first-pass draws are re-traced with the 113-draw table; results stay exact."""
    sc = ol.default_scene()
    mats = sc.materials.copy()
    mats[:, 4] = np.minimum(mats[:, 4], 0.3)     # low absorption: many rough hits per ray
    mats[:, 3] = np.maximum(mats[:, 3], 0.2)
    sc2 = ol.Scene(sc.origin, sc.dims, 16, sc.present, sc.res, sc.grid, mats)
    st = ol.make_settings(width=64, height=48, samples=1, max_bounces=5, lod_bounces=0.05)
    cam = camera_for(sc2, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    r = cam.render(0, want_rays=True)
    assert r.stats[9] > 0 and r.stats[10] == 0, r.stats
    o = ol.render(sc2, st, sc.cam_pos, sc.cam_rot, sc.cam_lens, r.pixels, libm=ol.LIBM_PORTABLE)
    got = active(r)
    assert got["counters"][:, 5].max() > 32
    for f in ("color", "alpha", "counters", "energy", "step", "life", "bounces", "pos", "vel"):
        assert np.array_equal(got[f], o["rays"][f]), f
    assert (r.stats[:8] == o["counters"]).all()
    # the camera now keeps 64 draws per seed (speed only): fewer re-traces, identical rays
    assert cam.fast_draws == 64
    cam.fast_draws = 32
    rf = check_frame_march(cam, o, 16, frame_march)   # (the pool's re-traces take their prefix counts off again)
    assert rf.stats[9] > 0
    r2 = cam.render(0, want_rays=True)
    assert 0 < r2.stats[9] < r.stats[9]
    for f in ("color", "alpha", "counters", "energy", "step", "life", "bounces", "pos", "vel"):
        assert np.array_equal(active(r2)[f], o["rays"][f]), f


# ------------------------------------------------------------------------------------------------- full sizes
def test_config2_full_frame_vs_oracle():
    """BASELINE config 2: mods/default, 1920x1080, samples 1, 4 bounces -- whole frame against the oracle."""
    sc = ol.default_scene()
    st = ol.make_settings(width=1920, height=1080, samples=1, max_bounces=4)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    r = cam.render(0, want_ray_rgba=True)
    o = ol.render(sc, st, sc.cam_pos, sc.cam_rot, sc.cam_lens, r.pixels, libm=ol.LIBM_PORTABLE, threads=16,
                  want_rays=False)
    assert np.array_equal(r.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32))
    assert (r.stats[:8] == o["counters"]).all() and r.stats[8] == 1920 * 1080
    trav = np.array(r.traversed(16), np.int64).reshape(-1, 3)
    assert np.array_equal(trav, o["traversed"])
    # tile partition property: the union of the 8 thread tiles is the full frame, bit for bit
    st8 = dict(st, threads=8)
    cam8 = camera_for(sc, settings_store(st8), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    full = r.image_u8.cpu().numpy()
    acc = np.zeros_like(full)
    for t in range(8):
        part = cam8.render(t, want_f32=False, want_traversed=False).image_u8.cpu().numpy()
        assert (acc[part.any(2)] == 0).all()
        acc |= part
    assert np.array_equal(acc, full)


def test_config3_sampled_pixels_vs_oracle():
    """BASELINE config 3: 3840x2160, samples 8, 8 bounces -- every 4th pixel in x and y (1/16 of the frame, 3.9 M rays)
    against the oracle, and the frame-level invariants on the whole frame."""
    sc = ol.default_scene()
    st = ol.make_settings(width=3840, height=2160, samples=8, max_bounces=8)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    r = cam.render(0, want_ray_rgba=True, want_image=True)
    assert r.stats[10] == 0
    xs, ys = np.meshgrid(np.arange(0, 3840, 4), np.arange(0, 2160, 4), indexing="ij")
    sub = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    o = ol.render(sc, st, sc.cam_pos, sc.cam_rot, sc.cam_lens, sub, libm=ol.LIBM_PORTABLE, threads=16, want_rays=True,
                  want_traversed=False)
    f32 = cam.tile_f32(0).cpu().numpy()
    assert np.array_equal(f32[sub[:, 1], sub[:, 0]], o["pix_mean"].astype(np.float32))
    # RGBA8: exact where the pixel has one sample; with more samples the byte is trunc(mean) by assumption
    # (Surface.set_at's float -> u8 conversion cannot be observed here: "parity unpinned")
    img8 = r.image_u8.cpu().numpy()
    assert np.array_equal(img8[sub[:, 1], sub[:, 0]], o["pix_rgba8"])
    # per-sample results of the sampled pixels
    rr = r.ray_rgba.cpu().numpy().view(np.uint32).reshape(-1, r.max_samples)
    # pixel index in the x-major list: x * H + y
    rows = rr[sub[:, 0].astype(np.int64) * 2160 + sub[:, 1]]
    import ctypes as C
    orc_st = ol._orc_settings(st)
    ns = np.array([int(ol.lib().orc_pixel_samples(C.byref(orc_st), int(x), int(y))) for x, y in sub.tolist()])
    exp = o["rays"]
    assert ns.sum() == len(exp)
    packed = (exp["color"][:, 0] | (exp["color"][:, 1] << 8) | (exp["color"][:, 2] << 16) |
              (exp["alpha"] << 24)).astype(np.uint32)
    want = np.zeros_like(rows)
    slot = np.arange(len(exp)) - np.repeat(np.cumsum(ns) - ns, ns)     # sample index of every oracle ray
    want[np.repeat(np.arange(len(sub)), ns), slot] = packed
    assert np.array_equal(rows, want)
    # ray count = sum of per-pixel sample counts (lod_edge trims samples: reference init.py:133-134)
    assert 60_000_000 < r.stats[8] < 66_355_200
    # a second render is bit-identical (static seeding: frame-invariant)
    r2 = cam.render(0)
    assert np.array_equal(r2.rgba_f32.cpu().numpy(), r.rgba_f32.cpu().numpy())
    assert (r2.stats == r.stats).all()


@pytest.mark.parametrize("partition", ["xor", "seed"])
def test_config4_eight_shards_equal_the_single_gpu_frame(partition):
    """BASELINE config 4 (3840x2160, samples 8, 8 bounces, pixels sharded over 8 ranks) on one GPU: the 8 shards of
    multigpu.rank_pixels -- the reference's (x ^ y) % 8 and the seed-class partition bench.py uses for N > 1 --
    rendered one after the other assemble to the single-shard frame bit for bit (RGBA8 SHA-256 as bench.py reports
    it), their event counters and ray counts add up to the full frame's, and the union of their traversed keys is
    the full frame's traversed set."""
    import hashlib
    import torch
    from python_raytracer_amd.multigpu import rank_pixels, rank_pixel_counts, merge_traversed
    W, H, S = 3840, 2160, 8
    sc = ol.default_scene()
    st = ol.make_settings(width=W, height=H, samples=S, max_bounces=8)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    full = cam.render(0, pixels=rank_pixels(W, H, 1, 0), want_f32=False)
    sha_full = hashlib.sha256(full.image_u8.cpu().numpy().tobytes()).hexdigest()
    image = torch.zeros_like(full.image_u8)
    total = np.zeros(9, np.int64)
    keys = []
    counts = rank_pixel_counts(W, H, 8, partition, S)
    assert counts.sum() == W * H
    for rank in range(8):
        px = rank_pixels(W, H, 8, rank, partition, S)
        assert len(px) == counts[rank]
        r = cam.render(0, pixels=px, want_f32=False)
        assert r.stats[10] == 0 and r.stats[11] == 0
        own = torch.from_numpy(px.astype(np.int64)).cuda()
        assert int((image[own[:, 1], own[:, 0]] != 0).sum()) == 0          # shards are disjoint
        image += r.image_u8                                                 # non-owned pixels of a tile are 0
        total += r.stats[:9].astype(np.int64)
        keys.append(r.traversed_keys)
        del r
    assert hashlib.sha256(image.cpu().numpy().tobytes()).hexdigest() == sha_full
    assert np.array_equal(total, full.stats[:9].astype(np.int64))
    merged = merge_traversed(keys)
    assert torch.equal(merged != -1, full.traversed_keys != -1)


# ------------------------------------------------------------------------------------------------- tile plan
@pytest.mark.one_march
def test_tile_plan_matches_numpy():
    """The static distinct-seed index (include/vrt.h, vrt_plan_build) against a numpy restatement."""
    import torch
    sc = ol.default_scene()
    st = ol.make_settings(width=200, height=120, samples=5, threads=3)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    for t in range(3):
        cam.render(t, want_image=False, want_f32=False, want_traversed=False)
        dp = cam._pixel_cache[t][1]
        px = dp.array.astype(np.int64)
        L = ol.lib()
        ost = ol._orc_settings(st)
        import ctypes as C
        ns = np.array([L.orc_pixel_samples(C.byref(ost), int(x), int(y)) for x, y in px])
        seeds = []
        for (x, y), n in zip(px, ns):
            seeds += [(1 + x) * (1 + y) * (1 + s) for s in range(n)]
        distinct = np.unique(np.array(seeds, np.int64))
        assert dp.n_distinct == len(distinct)
        raw = dp.plan.cpu().numpy()
        hdr = raw[:64].view(np.uint64)
        slots = len(px) * 5
        assert hdr[1] == len(px) and hdr[2] == slots and hdr[3] == len(distinct)
        assert hdr[6] == 0                                                      # a third of the window: not the full frame
        seed_list = raw[64:64 + 4 * slots].view(np.uint32)[: len(distinct)]
        assert np.array_equal(seed_list.astype(np.int64), distinct)            # sorted, unique
        off = 64 + ((4 * slots + 255) // 256) * 256
        idx = raw[off:off + 4 * slots].view(np.uint32).reshape(len(px), 5)
        for i in range(0, len(px), 37):
            for s in range(5):
                if s < ns[i]:
                    assert seed_list[idx[i, s]] == (1 + px[i, 0]) * (1 + px[i, 1]) * (1 + s)
                else:
                    assert idx[i, s] == 0xFFFFFFFF


def test_full_frame_resolve_equals_the_list_order_resolve():
    """resolve_kernel takes 16 x 16 tiles when the plan says the pixel list is the whole window in x-major order
    (PlanHeader.full_frame), one thread per list entry otherwise: same image, same means -- for window sizes that are
    and are not multiples of 16, and for the same pixels in another order (which must clear the flag)."""
    import torch
    sc = ol.default_scene()
    for w, h in ((64, 48), (70, 39), (16, 16), (33, 100)):
        st = ol.make_settings(width=w, height=h, samples=3)
        cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
        full = cam.render(0)
        dp = cam._pixel_cache[0][1]
        assert int(dp.plan[:64].cpu().numpy().view(np.uint64)[6]) == 1
        px = dp.array.copy()
        perm = np.random.default_rng(w * 1000 + h).permutation(len(px))
        dps = cam.upload_pixels(np.ascontiguousarray(px[perm]))
        shuf = cam.render(0, pixels=dps)
        assert int(dps.plan[:64].cpu().numpy().view(np.uint64)[6]) == 0
        assert torch.equal(full.image_u8, shuf.image_u8)
        assert torch.equal(full.rgba_f32[torch.from_numpy(perm).to(full.rgba_f32.device)], shuf.rgba_f32)
        o = ol.render(sc, st, sc.cam_pos, sc.cam_rot, sc.cam_lens, px, libm=ol.LIBM_PORTABLE)
        assert np.array_equal(full.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32))


# ------------------------------------------------------------------------------------------------- synthetic volume
@pytest.mark.one_march
def test_synthetic_volume_generator_and_render_512():
    """The on-device config-5 volume generator against the numpy generator (packed bytes identical), and a render of
    it against the oracle on sampled pixels (config-5 settings at 512^3 / 1024x1024 / 4 spp)."""
    import torch
    import ctypes as C
    from python_raytracer_amd import PackedScene, _native as nat
    n, cs = 512, 16
    mats = ol.default_scene().materials
    dsc = ol.synth_scene(n, mats)
    d = n // cs
    table = torch.zeros(d ** 3, dtype=torch.int32, device="cuda")
    vox = torch.zeros(n ** 3, dtype=torch.uint8, device="cuda")
    nat.check(nat.lib().vrt_synth_volume(n, cs, table.data_ptr(), vox.data_ptr(), None), "vrt_synth_volume")
    ref = PackedScene.from_dense(dsc.origin, dsc.dims, cs, dsc.present, dsc.res, dsc.grid, mats)
    assert np.array_equal(table.cpu().numpy().view(np.uint32), ref.chunk_table)
    assert np.array_equal(vox.cpu().numpy(), ref.voxels.reshape(-1))
    # 64^3 golden scene is the same generator: spot-check it through the fixture too
    g64 = ol.synth64_scene()
    assert np.array_equal(ol.synth_scene(64, mats).grid, g64.grid)
    st = ol.make_settings(width=1024, height=1024, samples=4, max_bounces=8, dist_max=512, dof=0.0, lod_edge=0.0,
                          lod_random=0.0, lod_samples=0.0, lod_bounces=0.0)
    from python_raytracer_amd import Camera
    from python_raytracer_amd.lib import vec3, quaternion
    cam = Camera(settings=settings_store(st))
    cam.set_packed_scene(PackedScene.from_device(dsc.origin, dsc.dims, cs, table, vox, d ** 3, mats, max_resolution=1))
    cam.pos, cam.rot = vec3(0.5, 0.5, 0.5), quaternion(0.0, 0.0, 0.0, 1.0)
    r = cam.render(0, want_ray_rgba=True)
    assert r.stats[10] == 0 and r.stats[11] == 0 and r.stats[8] == 1024 * 1024 * 4
    xs, ys = np.meshgrid(np.arange(5, 1024, 23), np.arange(3, 1024, 19), indexing="ij")
    sub = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    o = ol.render(dsc, st, [0.5, 0.5, 0.5], [0, 0, 0, 1], cam.lens, sub, libm=ol.LIBM_PORTABLE, threads=16,
                  want_traversed=False)
    f32 = cam.tile_f32(0).cpu().numpy()
    assert np.array_equal(f32[sub[:, 1], sub[:, 0]], o["pix_mean"].astype(np.float32))
    rr = r.ray_rgba.cpu().numpy().view(np.uint32).reshape(-1, 4)[sub[:, 0].astype(np.int64) * 1024 + sub[:, 1]]
    exp = o["rays"]
    packed = (exp["color"][:, 0] | (exp["color"][:, 1] << 8) | (exp["color"][:, 2] << 16) | (exp["alpha"] << 24))
    assert np.array_equal(rr.reshape(-1), packed.astype(np.uint32))
    # This dense world's chunk table (32^3 cells: not in LDS) is the identity, which the host detects and the march then
    # computes instead of reading (VRT_SCENE_TABLE_IS_IDENTITY); read from memory it must give the same frame.
    # (from_device checks the table on the device, once, where the scene is made; a table-order layout goes with it)
    assert cam._ensure_scene().table_identity()
    assert cam._c_scene(cam._ensure_scene()).flags == nat.SCENE_TABLE_IS_IDENTITY | nat.SCENE_LAYOUT_DENSE
    os.environ["VRT_TABLE_IDENTITY"] = "0"
    try:
        assert cam._c_scene(cam._ensure_scene()).flags == nat.SCENE_LAYOUT_DENSE
        r2 = cam.render(0, want_ray_rgba=True)
    finally:
        del os.environ["VRT_TABLE_IDENTITY"]
    assert torch.equal(r.ray_rgba, r2.ray_rgba) and (r.stats[:9] == r2.stats[:9]).all()
    assert torch.equal(r.traversed_keys, r2.traversed_keys)
    # a table that is not the identity must not be reported as one
    t2 = table.clone()
    t2[5] = 0
    holed = PackedScene.from_device(dsc.origin, dsc.dims, cs, t2, vox, d ** 3, mats, max_resolution=1)
    assert not holed.table_identity() and holed.dense       # (a hole: no identity, but every block where its cell is)
    t2[6] = 9 | (1 << 24)                                   # (a cell that names another cell's block: no table order)
    assert not PackedScene.from_device(dsc.origin, dsc.dims, cs, t2, vox, d ** 3, mats, max_resolution=1).dense


def test_config5_full_size_properties():
    """BASELINE config 5 at full size (1024^3 volume, 4096x4096, 16 spp, 8 bounces): ray count, no invalid rays,
    frame-to-frame determinism under static seeding, and the fp32 image being exact k/16 means of bytes."""
    import torch
    from python_raytracer_amd import Camera, PackedScene, _native as nat
    from python_raytracer_amd.lib import vec3, quaternion
    n, cs = 1024, 16
    d = n // cs
    mats = ol.default_scene().materials
    table = torch.zeros(d ** 3, dtype=torch.int32, device="cuda")
    vox = torch.zeros(n ** 3, dtype=torch.uint8, device="cuda")
    nat.check(nat.lib().vrt_synth_volume(n, cs, table.data_ptr(), vox.data_ptr(), None), "vrt_synth_volume")
    st = ol.make_settings(width=4096, height=4096, samples=16, max_bounces=8, dist_max=1024, dof=0.0, lod_edge=0.0,
                          lod_random=0.0, lod_samples=0.0, lod_bounces=0.0)
    cam = Camera(settings=settings_store(st))
    cam.set_packed_scene(PackedScene.from_device([-512] * 3, [d] * 3, cs, table, vox, d ** 3, mats, max_resolution=1))
    cam.pos, cam.rot = vec3(0.5, 0.5, 0.5), quaternion(0.0, 0.0, 0.0, 1.0)
    a = cam.render(0, want_traversed=True)
    assert a.stats[8] == 4096 * 4096 * 16 == 268435456 and a.stats[10] == 0 and a.stats[11] == 0
    b = cam.render(0, want_traversed=True)
    assert (a.stats[:9] == b.stats[:9]).all()
    fa, fb = a.rgba_f32, b.rgba_f32
    assert torch.equal(fa, fb) and torch.equal(a.image_u8, b.image_u8)
    assert torch.equal(fa * 16, torch.round(fa * 16))                       # sums of 16 integer samples
    assert torch.equal(a.traversed_keys, b.traversed_keys)
    # image-order RGBA8 equals the truncated compact fp32 means
    px = torch.from_numpy(a.pixels.astype(np.int64)).cuda()
    assert torch.equal(a.image_u8[px[:, 1], px[:, 0]].to(torch.float32), torch.floor(fa))


# ------------------------------------------------------------------------------------------------- randomised scenes
@pytest.mark.parametrize("seed", list(range(1, 25)) + [1099, 1192, 1287, 1344])  # 1099...: primary |vel|_inf > 1
def test_random_scenes_bit_exact(seed, frame_march):
    """Random sparse chunk layouts (missing chunks, resolutions 1..4, chunk sizes 8/16/32), random materials
    (incl. ior 0 / > 0.5 / < 0.5, zero roughness, emissive), random cameras and settings: every ray field bit-exact
    against the oracle."""
    rng = np.random.default_rng(seed)
    cs = int(rng.choice([8, 16, 32]))
    dims = rng.integers(1, 5, 3)
    origin = (rng.integers(-3, 2, 3) * cs).astype(np.int64)
    present = (rng.random(tuple(dims)) < 0.8).astype(np.uint8)
    if not present.any():
        present[0, 0, 0] = 1
    res = rng.integers(1, 5, tuple(dims)).astype(np.uint8)
    n_mat = int(rng.integers(1, 9))
    mats = np.zeros((n_mat, 7))
    mats[:, :3] = rng.integers(0, 256, (n_mat, 3))
    mats[:, 3] = rng.choice([0.0, 0.1, 0.5, 1.0], n_mat)                     # roughness
    mats[:, 4] = rng.choice([0.25, 0.5, 1.0, 1.5, 2.0], n_mat)               # absorption
    mats[:, 5] = rng.choice([0.0, 0.25, 0.5, 0.75, 1.0], n_mat)              # ior
    mats[:, 6] = rng.choice([0.0, 0.0, 0.5, 2.0], n_mat)                     # energy
    fill = rng.choice([0.02, 0.1, 0.4])
    grid = np.where(rng.random(tuple(dims * cs)) < fill, rng.integers(1, n_mat + 1, tuple(dims * cs)), 0).astype(np.uint8)
    sc = ol.Scene(origin, dims, cs, present, res, ol.Scene.camera_grid(grid, origin, dims, cs, present, res), mats)
    st = ol.make_settings(width=int(rng.integers(8, 40)), height=int(rng.integers(8, 40)), samples=int(rng.integers(1, 6)),
                          max_bounces=float(rng.choice([1, 2.5, 4, 8])), chunk_size=cs,
                          dist_max=int(rng.choice([16, 48, 96])), dist_min=int(rng.choice([0, 0, 2])),
                          dof=float(rng.choice([0.0, 0.5, 2.0])), lod_edge=float(rng.choice([0.0, 0.25, 0.9])),
                          lod_random=float(rng.choice([0.0, 0.25])), lod_samples=float(rng.choice([0.0, 0.5])),
                          lod_bounces=float(rng.choice([0.0, 0.5])), max_light=float(rng.choice([0.5, 1.0, 4.0])),
                          falloff=float(rng.choice([0.0, 0.25, 1.0])), shutter=float(rng.choice([0.0, 0.25])),
                          fov=float(rng.choice([60.0, 90.0, 150.0])))
    centre = origin + dims * cs / 2
    pos = centre + rng.uniform(-1, 1, 3) * dims * cs * 0.7
    if seed % 2:
        pos = np.round(pos)                                                 # integer camera: rays sit on voxel boundaries
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    lens = st["fov"] * np.pi / 8
    cam = camera_for(sc, settings_store(st), pos, q, lens)
    r = cam.render(0, want_rays=True)
    o = ol.render(sc, st, pos, q, lens, r.pixels, libm=ol.LIBM_PORTABLE)
    got, exp = active(r), o["rays"]
    assert len(got) == len(exp)
    for f in ("x", "y", "s", "color", "alpha", "counters", "ntrav", "detail", "energy", "step", "life", "bounces", "pos", "vel"):
        assert np.array_equal(got[f], exp[f]), (f, np.flatnonzero((got[f] != exp[f]).reshape(len(got), -1).any(1))[:5])
    assert np.array_equal(r.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32))
    assert np.array_equal(np.array(r.traversed(cs), np.int64).reshape(-1, 3), o["traversed"])
    assert (r.stats[:8] == o["counters"]).all()
    check_frame_march(cam, o, cs, frame_march)


@pytest.mark.parametrize("pos", [(0.0, 0.0, 0.0), (16.0, 16.0, 16.0), (8.0, 16.0, -16.0), (-0.0, 5.0, -5.0), (-16.0, 0.0, 31.0),
                                 (1e-300, -1e-300, 15.999999999999998), (32.0, -32.0, 0.5)])
def test_axis_aligned_rays_from_integer_and_boundary_cameras(pos, frame_march):
    """The integer forms of the march's box tests (floor by magic add, `floor(p) - chunk_min in [0, cs)` or `== cs` with p
    integral, the first-snap special case p == (0, 0, 0)) on the inputs that sit exactly on their edges: unrotated camera,
    no jitter, dist_min 0, even image size -- the centre column / row rays move along the axes planes with coordinates that
    stay integers, the camera sits on chunk corners, faces, the origin and one ulp beside them.  Both kernels (the
    recording one and the frame march) against the oracle, resolutions 1..3."""
    rng = np.random.default_rng(12345)
    cs = 16
    dims = np.array([4, 4, 4])
    origin = np.array([-32, -32, -32], np.int64)
    present = (rng.random(tuple(dims)) < 0.85).astype(np.uint8)
    res = rng.integers(1, 4, tuple(dims)).astype(np.uint8)
    mats = np.array([[200, 40, 40, 0.0, 0.5, 0.0, 0.0], [40, 200, 40, 0.5, 1.0, 0.75, 0.0], [40, 40, 200, 0.1, 0.25, 0.25, 0.5],
                     [220, 220, 220, 1.0, 2.0, 1.0, 0.0]])
    grid = np.where(rng.random(tuple(dims * cs)) < 0.08, rng.integers(1, 5, tuple(dims * cs)), 0).astype(np.uint8)
    for rm in (1, 3):                                   # resolutions <= 1 only (RESMODE 0 kernel) and up to 3
        r_ = np.minimum(res, rm).astype(np.uint8)
        sc = ol.Scene(origin, dims, cs, present, r_, ol.Scene.camera_grid(grid, origin, dims, cs, present, r_), mats)
        st = ol.make_settings(width=32, height=24, samples=2, max_bounces=4.0, chunk_size=cs, dist_max=96, dist_min=0,
                              dof=0.0, lod_edge=0.0, lod_random=0.0, lod_samples=0.0, fov=90.0)
        q = np.array([0.0, 0.0, 0.0, 1.0])
        lens = st["fov"] * np.pi / 8
        cam = camera_for(sc, settings_store(st), np.array(pos), q, lens)
        r = cam.render(0, want_rays=True)
        o = ol.render(sc, st, np.array(pos), q, lens, r.pixels, libm=ol.LIBM_PORTABLE)
        got, exp = active(r), o["rays"]
        for f in ("color", "alpha", "counters", "ntrav", "energy", "step", "life", "bounces", "pos", "vel"):
            assert np.array_equal(got[f], exp[f]), (rm, f, np.flatnonzero((got[f] != exp[f]).reshape(len(got), -1).any(1))[:5])
        assert np.array_equal(np.array(r.traversed(cs), np.int64).reshape(-1, 3), o["traversed"])
        check_frame_march(cam, o, cs, frame_march)    # the frame march (no ray records)


# ------------------------------------------------------------------------------------------------- chunk selection
def test_chunk_update_culling_sequence_vs_reference():
    """Camera.chunk_update (vrt_select_chunks): LOD selection + culling feedback over five consecutive frames of the
    real reference with culling on (tests/golden/culling_sequence.npz), rendering each frame over the selected chunks."""
    import json as _json
    from python_raytracer_amd import Camera, PackedScene
    from python_raytracer_amd.lib import vec3, quaternion
    z = np.load(os.path.join(ol.GOLDEN, "culling_sequence.npz"))
    sc = ol.default_scene()
    s = _json.loads(bytes(z["settings"]).decode())
    st = ol.make_settings(**{k: s[k] for k in ol.DEFAULT_SETTINGS if k in s})
    sst = settings_store(st)
    sst.culling = True
    cam = Camera(settings=sst)
    world = PackedScene.from_dense(sc.origin, sc.dims, 16, sc.present, np.ones_like(sc.res), sc.grid_lod0, sc.materials)
    cam.set_world_scene(world)
    cam.rot = quaternion(*[float(v) for v in z["cam_rot"]])
    cam.lens = float(z["cam_lens"][0])
    prev = None
    for it in range(5):
        cam.pos = vec3(*[float(v) for v in z["pos_%d" % it]])
        # alternate between the device-side keys of the previous result and the list tile() returns
        feed = prev if it % 2 == 0 else (prev.traversed(16) if prev is not None else None)
        table = cam.chunk_update(feed).cpu().numpy().view(np.uint32).reshape(tuple(sc.dims))
        assert np.array_equal((table != 0).astype(np.uint8), z["present_%d" % it]), it
        assert np.array_equal((table >> 24).astype(np.uint8), z["res_%d" % it]), it
        r = cam.render(0)
        px = r.pixels
        assert np.array_equal(r.rgba_f32.cpu().numpy(), z["pix_%d" % it][px[:, 1], px[:, 0]].astype(np.float32)), it
        assert np.array_equal(np.array(r.traversed(16)).reshape(-1, 3), z["traversed_%d" % it]), it
        prev = r
    # culling off: every chunk with voxels, LODs of the fixture scene
    sst.culling = False
    cam.pos = vec3(*[float(v) for v in sc.cam_pos])
    table = cam.chunk_update(None).cpu().numpy().view(np.uint32).reshape(tuple(sc.dims))
    assert np.array_equal((table != 0).astype(np.uint8), sc.present) and np.array_equal((table >> 24).astype(np.uint8), sc.res)
    # random cameras against the oracle's restatement
    rng = np.random.default_rng(3)
    sst.culling = True
    for _ in range(5):
        pos = rng.uniform(-150, 150, 3)
        trav = (rng.integers(-6, 6, (40, 3)) * 16).astype(np.float64)
        cam.pos = vec3(*pos)
        table = cam.chunk_update([tuple(t) for t in trav]).cpu().numpy().view(np.uint32).reshape(tuple(sc.dims))
        pres, res = ol.select_chunks(sc.origin, sc.dims, 16, sc.present, pos, 192, 2, True, trav)
        assert np.array_equal((table != 0).astype(np.uint8), pres) and np.array_equal((table >> 24).astype(np.uint8), res)


def test_camera_moves_closer_without_reselecting():
    """The camera chunk table outlives camera moves (the reference gates Window.chunk_update by chunk_rate but moves
    cam.pos every frame, init.py:391, 464): a table selected far away holds resolution-2 and -3 chunks; after the camera
    has moved close -- where a fresh selection would hold resolution 1 only -- the frame must still be marched with the
    snapping those resolutions need.  Against the oracle over the very table the selection wrote."""
    from python_raytracer_amd import Camera, PackedScene
    from python_raytracer_amd.lib import vec3, quaternion
    sc = ol.default_scene()
    st = ol.make_settings(width=96, height=54, samples=2, max_bounces=4, dist_max=192, chunk_lod=2)
    sst = settings_store(st)
    sst.culling = False
    cam = Camera(settings=sst)
    world = PackedScene.from_dense(sc.origin, sc.dims, 16, sc.present, np.ones_like(sc.res), sc.grid_lod0, sc.materials)
    cam.set_world_scene(world)
    cam.rot = quaternion(*[float(v) for v in sc.cam_rot])
    cam.lens = float(sc.cam_lens)
    far = np.array([-12.0, 4.0, 150.0])
    cam.pos = vec3(*far)
    table = cam.chunk_update(None).cpu().numpy().view(np.uint32).reshape(tuple(sc.dims))
    res_far = (table >> 24).astype(np.uint8)
    assert int(res_far.max()) == 3
    near = np.array([float(v) for v in sc.cam_pos])
    cam.pos = vec3(*near)                                 # ... and no chunk_update()
    assert cam._max_selected_resolution(world) < 3        # (a bound taken now would understate the table's resolutions)
    assert cam._c_scene(cam._ensure_scene()).max_resolution == 3
    pres = (table != 0).astype(np.uint8)
    osc = ol.Scene(sc.origin, sc.dims, 16, pres, res_far,
                   ol.Scene.camera_grid(sc.grid_lod0, sc.origin, sc.dims, 16, pres, res_far), sc.materials)
    r = cam.render(0, want_ray_rgba=True)
    o = ol.render(osc, st, near, sc.cam_rot, sc.cam_lens, r.pixels, libm=ol.LIBM_PORTABLE)
    assert np.array_equal(r.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32))
    assert (r.stats[:8] == o["counters"]).all()
    assert np.array_equal(np.array(r.traversed(16), np.int64).reshape(-1, 3), o["traversed"])


# ------------------------------------------------------------------------------------------------- scheduling knobs
_KNOB_SCRIPT = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import oracle_lib as ol
from gpu_util import camera_for, settings_store
sc = ol.default_scene()
st = ol.make_settings(width=160, height=90, samples=4, max_bounces=8)
cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
h = hashlib.sha256()
r = cam.render(0, want_rays=True)                       # the record-keeping kernel (generic resolutions)
rays = r.rays[r.rays["s"] >= 0]
for f in ("color", "alpha", "counters", "ntrav", "energy", "step", "life", "bounces", "pos", "vel"):
    h.update(np.ascontiguousarray(rays[f]).tobytes())
for r in (r, cam.render(0, want_ray_rgba=True)):        # ... and the fast kernel a frame normally uses
    h.update(r.rgba_f32.cpu().numpy().tobytes()); h.update(r.image_u8.cpu().numpy().tobytes())
    h.update(np.array(r.traversed(16)).tobytes()); h.update(r.stats[:9].tobytes())
h.update(r.ray_rgba.cpu().numpy().tobytes())
cam.cache_draws = False                                 # ... and without the cached tables (VRT_FUSE_RAYGEN: who makes the ray records)
r = cam.render(0, want_ray_rgba=True)
h.update(r.rgba_f32.cpu().numpy().tobytes()); h.update(r.ray_rgba.cpu().numpy().tobytes()); h.update(r.stats[:9].tobytes())
print("HASH", h.hexdigest())
"""


@pytest.mark.one_march
def test_scheduling_knobs_do_not_change_results():
    """Wave count, hand-out chunk size, rays per launch, the slow-body thresholds, the LDS shortcuts, the kernel variant
    (speculation depth, resolution mode) and the lookup variants (material bytes / occupancy words in registers / 8^3
    occupancy bricks staged in LDS) and the ray pool (march_pool_kernel: rays regrouped between lanes through LDS, against
    march_kernel's one ray per lane) only schedule work or fetch the same information another way: every output (rays,
    image, per-sample results, traversed order, counters) must be bit-identical for all of them (each setting runs in
    its own process because the knobs are read once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _KNOB_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    hashes = {}
    for env in ({}, {"VRT_T_HIT": "1", "VRT_T_END": "1"}, {"VRT_T_HIT": "64", "VRT_T_END": "64"}, {"VRT_CHUNK": "0"},
                {"VRT_CHUNK": "64", "VRT_MARCH_GRID": "7"}, {"VRT_MARCH_GRID": "1", "VRT_T_HIT": "17", "VRT_MAX_ITERS": "1"},
                {"VRT_T_END": "5", "VRT_T_HIT": "9", "VRT_MAX_ITERS": "50"}, {"VRT_BATCH_LOG2": "13"}, {"VRT_BATCH_LOG2": "24"},
                {"VRT_POW_MEMO": "frame"}, {"VRT_SPEC_DEEP": "1"}, {"VRT_SPEC_DEEP": "0"}, {"VRT_TRAV_LDS": "0"}, {"VRT_RESMODE": "2"},
                {"VRT_LOOKUP": "1"}, {"VRT_LOOKUP": "2"}, {"VRT_LOOKUP": "1", "VRT_SPEC_DEEP": "1"},
                {"VRT_LOOKUP": "2", "VRT_SPEC_DEEP": "1", "VRT_T_HIT": "3"}, {"VRT_POOL": "0"}, {"VRT_POOL": "1", "VRT_POOL_MIN_RAYS": "0"},
                {"VRT_POOL": "0", "VRT_T_HIT": "1", "VRT_T_END": "1"}, {"VRT_POOL": "0", "VRT_CHUNK": "0", "VRT_MARCH_GRID": "3"},
                {"VRT_POOL": "0", "VRT_SPEC_DEEP": "0", "VRT_TRAV_LDS": "0"}, {"VRT_POOL": "0", "VRT_RESMODE": "2"},
                {"VRT_WADDR": "1"}, {"VRT_WADDR": "1", "VRT_POOL": "0"}, {"VRT_WADDR": "1", "VRT_DENSE": "0"},
                {"VRT_TRAV_LDS": "0", "VRT_DEFER_VISIT": "2"}, {"VRT_TRAV_LDS": "0", "VRT_DEFER_VISIT": "2", "VRT_POOL": "0"},
                {"VRT_FUSE_RAYGEN": "0"}, {"VRT_TRAV_WINDOW": "2"}, {"VRT_TRAV_WINDOW": "2", "VRT_POOL": "0"}, {"VRT_TRAV_WINDOW": "0"},
                {"VRT_TRAV_WINDOW": "2", "VRT_DEFER_VISIT": "2"}, {"VRT_TRAV_WINDOW": "2", "VRT_DEFER_VISIT": "2", "VRT_POOL": "0"},
                {"VRT_WADDR": "1", "VRT_POOL": "1", "VRT_POOL_MIN_RAYS": "0", "VRT_TRAV_LDS": "0", "VRT_CHUNK": "64"},
                {"VRT_POOL": "1", "VRT_POOL_MIN_RAYS": "0", "VRT_POOL_T_HIT": "1", "VRT_POOL_T_END": "1", "VRT_POOL_SWAP_MIN": "1", "VRT_POOL_REFILL_MIN": "1", "VRT_POOL_KEEP": "1", "VRT_POOL_ITERS": "9"},
                {"VRT_POOL": "1", "VRT_POOL_MIN_RAYS": "0", "VRT_POOL_T_HIT": "112", "VRT_POOL_T_END": "112", "VRT_CHUNK": "64"},
                {"VRT_POOL": "1", "VRT_POOL_MIN_RAYS": "0", "VRT_POOL_T_HIT": "64", "VRT_POOL_T_END": "7", "VRT_POOL_SWAP_MIN": "64", "VRT_MARCH_GRID": "2", "VRT_POOL_KEEP": "64"},
                {"VRT_POOL": "1", "VRT_POOL_MIN_RAYS": "0", "VRT_POOL_T_HIT": "9", "VRT_POOL_T_END": "100", "VRT_POOL_REFILL_MIN": "64", "VRT_SPEC_DEEP": "0",
                 "VRT_TRAV_LDS": "0"}):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", script], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (env, out.stderr[-2000:])
        hashes[str(env)] = [l for l in out.stdout.splitlines() if l.startswith("HASH")][0]
    assert len(set(hashes.values())) == 1, hashes


def test_traversed_keys_reset_by_the_call_or_kept_for_the_caller():
    """vrt_traversed.reset: Camera.render lets vrt_render_tile set the keys to "never visited" itself (one launch less per
    frame); with reset = 0 the keys are the caller's -- what was in them takes part in the minimum (several tiles into one box)."""
    import torch
    sc = ol.default_scene()
    st = ol.make_settings(width=96, height=54, samples=2, max_bounces=4)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    want = cam.render(0)
    visited = (want.traversed_keys != -1).nonzero().flatten()
    assert len(visited) > 4
    box = cam._trav_box

    def callers_keys(want_traversed):
        tr, keys = box(want_traversed)
        assert tr.reset == 1
        keys.fill_(-1)
        keys[visited[0]] = 0            # an earlier tile's visit: smaller than any key of this frame's rays but ray 0's first
        keys[visited[1]] = 1 << 62      # ... and one larger than any of them
        tr.reset = 0
        return tr, keys
    cam._trav_box = callers_keys
    got = cam.render(0)
    expect = want.traversed_keys.clone()
    expect[visited[0]] = 0
    assert torch.equal(got.traversed_keys, expect)


@pytest.mark.parametrize("size", [(96, 64, 4), (64, 96, 4), (160, 32, 4), (37, 64, 8), (256, 128, 2)])
def test_tiled_hand_out_gives_the_same_frame(frame_march, size):
    """A measured variant of march_pool_kernel (VRT_TILED=1: scenes beyond the caches) hands the whole window's rays out as square
    pixel tiles in Morton order, an eighth of the (padded) tile grid per XCD from a head of its own, instead of in list order
    (tile_ticket): scheduling only.  VRT_TILED=2 puts it to work on the default scene (with the kernel instances such scenes get:
    VRT_TRAV_LDS=0 VRT_DEFER_VISIT=2); windows that are not a power of two of tiles across or down leave tickets of the
    padding to skip, and heads that run dry at different times.  Every output equals the frame in list order and the oracle's."""
    if not frame_march.startswith("pool") or frame_march.endswith("-ahead"):
        pytest.skip("the ray pool's hand-out (the look-ahead variant has no such instance)")
    w, h, spp = size
    sc = ol.default_scene()
    st = ol.make_settings(width=w, height=h, samples=spp, max_bounces=4)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    base = cam.render(0, want_ray_rgba=True)
    assert int(base.stats[12]) >> 32 == 0
    o = ol.render(sc, st, sc.cam_pos, sc.cam_rot, sc.cam_lens, base.pixels, libm=ol.LIBM_PORTABLE, want_rays=False)
    os.environ["VRT_TRAV_LDS"], os.environ["VRT_DEFER_VISIT"], os.environ["VRT_TILED"] = "0", "2", "2"
    try:
        r = cam.render(0, want_ray_rgba=True)
    finally:
        del os.environ["VRT_TRAV_LDS"], os.environ["VRT_DEFER_VISIT"], os.environ["VRT_TILED"]
    for got in (r,):
        assert int(got.stats[12]) >> 32 == int(got.stats[12]) & 0xffffffff > 0, got.stats     # every workgroup took tiles
        assert np.array_equal(got.ray_rgba.cpu().numpy(), base.ray_rgba.cpu().numpy()) and (got.stats[:9] == base.stats[:9]).all()
        assert np.array_equal(got.traversed_keys.cpu().numpy(), base.traversed_keys.cpu().numpy())
        assert np.array_equal(got.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32)) and (got.stats[:8] == o["counters"]).all()


@pytest.mark.gpu
def test_frames_on_two_streams_equal_sequential_frames():
    """Frames submitted on different streams (bench.py --frames-in-flight: frame k + 1 starts while frame k's last waves
    drain) use one workspace per stream and share the read-only tables: every output equals the same frame rendered
    alone."""
    import torch
    from python_raytracer_amd.lib import vec3
    sc = ol.default_scene()
    st = ol.make_settings(width=160, height=90, samples=4, max_bounces=8)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    poses = [sc.cam_pos + np.array([0.5 * k, 0.1 * k, -0.25 * k]) for k in range(6)]
    alone = []
    for pos in poses:
        cam.pos = vec3(*[float(v) for v in pos])
        r = cam.render(0, want_ray_rgba=True)
        alone.append((r.rgba_f32.clone(), r.image_u8.clone(), r.ray_rgba.clone(), r.traversed_keys.clone(), r.stats.copy()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = []
    for k, pos in enumerate(poses):
        cam.pos = vec3(*[float(v) for v in pos])
        with torch.cuda.stream(streams[k % 2]):
            got.append(cam.render(0, want_ray_rgba=True, check=False))
    torch.cuda.synchronize()
    assert len(cam._workspace) >= 3                      # the default stream's and one per side stream
    for (f32, img, rr, keys, stats), r in zip(alone, got):
        assert torch.equal(f32, r.rgba_f32) and torch.equal(img, r.image_u8) and torch.equal(rr, r.ray_rgba)
        assert torch.equal(keys, r.traversed_keys) and (stats == r._stats_dev.cpu().numpy()).all()


@pytest.mark.gpu
def test_world_flow_runs_the_resolution_2_kernel_and_renders_the_fixture():
    """The reference's own flow -- whole world resident, Camera.chunk_update selects chunks and LOD per frame (init.py:
    441-452, chunk_lod = 2) -- must (a) render exactly what the pre-selected fixture renders and (b) tell the library
    max_resolution = 2, because no chunk of the default world is far enough from the default camera for LOD 2: the frame
    then runs the resolution <= 2 kernel (8-step speculation) instead of the generic one a bare chunk_lod + 1 selects."""
    import torch
    import bench
    from python_raytracer_amd import Camera
    from python_raytracer_amd.data import make_settings
    from python_raytracer_amd.lib import vec3, quaternion
    st = make_settings(width=192, height=108, samples=4, max_bounces=8.0, threads=1)
    st.culling = False
    frames = []
    for world in (False, True):
        scene, cam_pos, cam_rot, _ = bench.load_default_scene(world=world)
        cam = Camera(settings=st)
        cam.pos, cam.rot = vec3(*cam_pos.tolist()), quaternion(*cam_rot.tolist())
        if world:
            cam.set_world_scene(scene)
            cam.chunk_update(None)
            assert int(st.chunk_lod) == 2 and cam._c_scene(cam._ensure_scene()).max_resolution == 2
            table = cam._camera_table.cpu().numpy().view(np.uint32)
            assert int((table >> 24).max()) == 2                 # ... and the bound is tight here
        else:
            cam.set_packed_scene(scene)
        r = cam.render(0, want_ray_rgba=True)
        frames.append((r.rgba_f32.clone(), r.ray_rgba.clone(), r.stats.copy(), np.array(r.traversed(16))))
    assert torch.equal(frames[0][0], frames[1][0]) and torch.equal(frames[0][1], frames[1][1])
    assert (frames[0][2][:12] == frames[1][2][:12]).all() and np.array_equal(frames[0][3], frames[1][3])
    # a camera far outside the world: every chunk beyond 2/3 of dist_max gets LOD 2 -> resolution 3 -> the generic kernel
    cam.pos = vec3(-12.0, 4.0, 190.0)
    cam.chunk_update(None)
    assert cam._c_scene(cam._ensure_scene()).max_resolution == 3
    table = cam._camera_table.cpu().numpy().view(np.uint32)
    assert int((table >> 24).max()) == 3


@pytest.mark.gpu
def test_first_frames_on_different_streams_wait_for_the_table_builds():
    """The cached draw and ray tables are built asynchronously on whichever stream renders first.  A camera whose very
    first frame runs on side stream A and whose second frame runs at once on side stream B (no synchronisation in
    between) must still read finished tables on B: Camera orders every stream behind the builds with an event."""
    import torch
    from python_raytracer_amd.lib import vec3
    sc = ol.default_scene()
    st = ol.make_settings(width=320, height=180, samples=8, max_bounces=8)
    ref_cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    want = []
    for k in range(2):
        ref_cam.pos = vec3(*[float(v) for v in sc.cam_pos + np.array([0.25 * k, 0.0, -0.5 * k])])
        r = ref_cam.render(0, want_ray_rgba=True)
        want.append((r.rgba_f32.clone(), r.ray_rgba.clone(), r.stats.copy()))
    torch.cuda.synchronize()
    for trial in range(3):
        cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)   # nothing built yet
        a, b = torch.cuda.Stream(), torch.cuda.Stream()
        got = []
        for k, stream in enumerate((a, b)):
            cam.pos = vec3(*[float(v) for v in sc.cam_pos + np.array([0.25 * k, 0.0, -0.5 * k])])
            with torch.cuda.stream(stream):
                got.append(cam.render(0, want_ray_rgba=True, check=False))
        torch.cuda.synchronize()
        for (f32, rr, stats), r in zip(want, got):
            assert torch.equal(f32, r.rgba_f32) and torch.equal(rr, r.ray_rgba)
            assert (stats == r._stats_dev.cpu().numpy()).all()


@pytest.mark.gpu
def test_cached_tables_give_identical_frames(frame_march):
    """Camera.cache_draws (the default): with static seeds the draw table and the ray table (lens quaternion + life per
    ray slot) are built once (vrt_draw_table_build, vrt_ray_table_build) and reused; every output must equal the render
    that re-seeds and regenerates both in every frame -- also after the camera moved and turned, after the lens changed
    (the ray table depends on it) and after the table width changed."""
    import torch
    sc = ol.default_scene()
    st = ol.make_settings(width=128, height=72, samples=4, max_bounces=8)
    a = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    b = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    assert b.cache_draws is True
    a.cache_draws = False
    from python_raytracer_amd.lib import vec3, quaternion
    moves = [sc.cam_pos, sc.cam_pos + np.array([1.5, 0.25, -2.0]), sc.cam_pos, sc.cam_pos, sc.cam_pos]
    tables = []
    for frame, pos in enumerate(moves):
        a.pos = vec3(*[float(v) for v in pos])
        b.pos = vec3(*[float(v) for v in pos])
        if frame == 1:
            a.rot = b.rot = quaternion(0.05, 0.3, -0.1, 0.94)
        if frame == 2:
            a.fast_draws = b.fast_draws = 64
        if frame == 3:
            a.lens = b.lens = float(sc.cam_lens) * 0.8
        ra, rb = a.render(0, want_ray_rgba=True), b.render(0, want_ray_rgba=True)
        # (a has no ray table: its march derives the records itself -- unless it is the look-ahead variant, which has no such instance)
        assert int(rb.stats[15]) == 0 and (int(ra.stats[15]) > 0) == (not frame_march.endswith("-ahead")), (ra.stats, rb.stats)
        assert torch.equal(ra.rgba_f32, rb.rgba_f32) and torch.equal(ra.image_u8, rb.image_u8)
        assert torch.equal(ra.ray_rgba, rb.ray_rgba)
        assert (ra.stats[:15] == rb.stats[:15]).all() and ra.traversed(16) == rb.traversed(16)
        dp = b._pixels_tensor(0, None)
        tables.append((dp.draw_table.data_ptr(), dp.ray_table.data_ptr()))
    dp = b._pixels_tensor(0, None)
    assert dp.draw_table is not None and dp.draw_key[1] == 64 and dp.ray_table is not None
    assert tables[0] == tables[1]                       # moving / turning the camera rebuilds nothing
    assert tables[3][0] == tables[2][0] and tables[4] == tables[3]   # a new lens keeps the draws
    assert a._pixels_tensor(0, None).draw_table is None
    # a non-static run never uses the cache (its nonce changes every frame)
    st2 = settings_store(st)
    st2.static = False
    c = camera_for(sc, st2, sc.cam_pos, sc.cam_rot, sc.cam_lens)
    c.render(0)
    assert c._pixels_tensor(0, None).draw_table is None


@pytest.mark.one_march
@pytest.mark.gpu
def test_bench_two_ranks_render_the_single_gpu_frame():
    """bench.py end to end: one rank, and two ranks launched exactly as the driver does (torch.distributed.run; gloo
    stands in for RCCL because this box has one GPU, both ranks share it) with either pixel partition.  The gathered
    RGBA8 frame must be bit-identical (same SHA-256) and the whole-job ray counts equal."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(n, extra):
        env = dict(os.environ)
        env["VRT_BENCH_BACKEND"] = "gloo"
        cmd = [sys.executable]
        if n > 1:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                    "--master-port", str(port)]
        cmd += [os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--config", "c2",
                "--no-cpu"] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        return json.loads(lines[0])

    one = run(1, [])
    assert one["n_gpus"] == 1 and one["config"]["primary_rays"] == 1920 * 1080
    for key in ("roofline", "kernel_ms_per_step", "reseeded_every_frame"):
        assert key in one
    for extra in ([], ["--partition", "xor"]):
        two = run(2, extra)
        assert two["n_gpus"] == 2 and two["scaling"] == "strong"
        assert two["config"]["image_sha256"] == one["config"]["image_sha256"]
        assert two["config"]["primary_rays"] == one["config"]["primary_rays"]
        assert two["config"]["bounce_rays"] == one["config"]["bounce_rays"]
        # the line checks itself against the committed single-GPU line of the configuration (profiles/)
        assert two["matches_single_gpu"] is True and two["single_gpu_reference"]["image_sha256"] == one["config"]["image_sha256"]


@pytest.mark.one_march
@pytest.mark.gpu
def test_bench_two_ranks_over_rccl():
    """The same through RCCL (torch.distributed backend "nccl"), one rank per GPU, launched as the driver launches it:
    needs two GPUs, so it is skipped on a one-GPU box.  The line must report that it matches the committed single-GPU
    frame of config 3 (image hash + whole-job ray counts), and bench.py exits non-zero otherwise."""
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.pop("VRT_BENCH_BACKEND", None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["matches_single_gpu"] is True


@pytest.mark.gpu
def test_frame_is_graph_capturable():
    """vrt_render_tile neither allocates nor synchronises: a whole frame (seeding, ray generation, march, retrace tiers,
    resolve) can be captured into a HIP graph and replayed, also after the camera moved (the camera is passed by
    value, so a graph holds the camera it was captured with)."""
    import torch
    sc = ol.default_scene()
    st = ol.make_settings(width=160, height=90, samples=2, max_bounces=4)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    dp = cam.upload_pixels(np.concatenate(ol.pixel_lists(160, 90, 1)))
    ref = cam.render(0, pixels=dp)                       # also warms the plan, the workspace and the pow memo
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cam.render(0, pixels=dp, check=False)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r = cam.render(0, pixels=dp, check=False)
    for _ in range(3):
        r.rgba_f32.zero_()
        r.image_u8.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(r.rgba_f32, ref.rgba_f32) and torch.equal(r.image_u8, ref.image_u8)
        assert (r._stats_dev.cpu().numpy()[:9] == ref.stats[:9]).all()


@pytest.mark.gpu
def test_release_caches_then_render_again():
    """vrt_release_caches frees the per-(device, falloff) pow tables (after synchronising their device); a frame rendered
    without one (it memoises into its workspace), and the next one after vrt_pow_memo_create, are identical."""
    import torch
    import python_raytracer_amd as pra
    from python_raytracer_amd import _native as nat
    sc = ol.default_scene()
    st = ol.make_settings(width=96, height=54, samples=2, max_bounces=4)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    a = cam.render(0)
    assert nat.lib().vrt_release_caches() == 0          # no synchronisation by the caller needed
    b = cam.render(0)                                    # the camera still believes the table exists: per-frame memo
    assert torch.equal(a.rgba_f32, b.rgba_f32) and (a.stats[:9] == b.stats[:9]).all()
    pra.release_caches()
    c = cam.render(0)                                    # re-created by the wrapper
    assert torch.equal(a.rgba_f32, c.rgba_f32) and (a.stats[:9] == c.stats[:9]).all()
