"""Soak (GPU box): random scenes with wider ranges than tests/test_gpu_parity.py::test_random_scenes_bit_exact --
every ray field, image, traversed list and counters against the oracle, then the explicit-ray entry point
(Camera.trace_many) against the tile's own ray records.  usage: soak_scenes.py FIRST_SEED LAST_SEED
The frame kernel under test follows the environment like everywhere: VRT_POOL / VRT_POOL_MIN_RAYS=0 (ray pool on these tiny
launches), VRT_WADDR=1 (look-ahead across chunk borders; SOAK_MAXRES=2 keeps the scenes to the resolutions it exists for).
SOAK_UNCACHED=1: the frame kernel's render runs without cached tables (Camera.cache_draws = False): with SOAK_MAXRES=2 its
lanes then make their own ray records -- lenses up to 179 degrees put the joint sin / cos block's fallback to work too."""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib as ol
from gpu_util import camera_for, settings_store
from python_raytracer_amd import _native as nat

def active(r):
    return r.rays[r.rays["s"] >= 0]

def got_slots(r):
    """Indices of the active ray slots in the tile's slot order."""
    return np.flatnonzero(r.rays["s"] >= 0)

def one(seed):
    rng = np.random.default_rng(seed)
    cs = int(rng.choice([8, 16, 32, 64]))
    dims = rng.integers(1, 4, 3)
    far = int(rng.choice([0, 0, 1000, -50000, 1 << 20]))
    origin = ((rng.integers(-3, 2, 3) + far) * cs).astype(np.int64)
    present = (rng.random(tuple(dims)) < 0.8).astype(np.uint8)
    if not present.any(): present[0, 0, 0] = 1
    res = rng.integers(1, int(rng.choice([3, 5, 10])), tuple(dims)).astype(np.uint8)
    if os.environ.get("SOAK_MAXRES"):   # e.g. 2: only scenes the resolution <= 2 kernels (and march_step_w) run
        res = np.minimum(res, int(os.environ["SOAK_MAXRES"])).astype(np.uint8)
    n_mat = int(rng.integers(1, 20))
    mats = np.zeros((n_mat, 7))
    mats[:, :3] = rng.integers(0, 256, (n_mat, 3))
    mats[:, 3] = rng.choice([0.0, 0.1, 0.5, 1.0, 2.5], n_mat)
    mats[:, 4] = rng.choice([0.05, 0.25, 0.5, 1.0, 1.5, 2.0, 7.0], n_mat)
    mats[:, 5] = rng.choice([0.0, 0.25, 0.5, 0.75, 1.0], n_mat)
    mats[:, 6] = rng.choice([0.0, 0.0, 0.5, 2.0], n_mat)
    fill = rng.choice([0.005, 0.02, 0.1, 0.4, 0.9])
    grid = np.where(rng.random(tuple(dims * cs)) < fill, rng.integers(1, n_mat + 1, tuple(dims * cs)), 0).astype(np.uint8)
    sc = ol.Scene(origin, dims, cs, present, res, ol.Scene.camera_grid(grid, origin, dims, cs, present, res), mats)
    st = ol.make_settings(width=int(rng.integers(1, 48)), height=int(rng.integers(1, 48)), samples=int(rng.integers(1, 10)),
                          max_bounces=float(rng.choice([0.5, 1, 2.5, 4, 8, 16])), chunk_size=cs,
                          dist_max=int(rng.choice([4, 16, 48, 96, 200])), dist_min=int(rng.choice([0, 0, 2, 3])),
                          dof=float(rng.choice([0.0, 0.5, 2.0, 10.0])), lod_edge=float(rng.choice([0.0, 0.25, 0.9, 1.0])),
                          lod_random=float(rng.choice([0.0, 0.25, 1.0])), lod_samples=float(rng.choice([0.0, 0.5, 3.0])),
                          lod_bounces=float(rng.choice([0.0, 0.5, 2.0])), max_light=float(rng.choice([0.1, 0.5, 1.0, 4.0])),
                          falloff=float(rng.choice([0.0, 0.25, 1.0, 3.0])), shutter=float(rng.choice([0.0, 0.25, 1.0])),
                          fov=float(rng.choice([20.0, 60.0, 90.0, 150.0, 179.0])))
    if st["dist_min"] >= st["dist_max"]: st["dist_min"] = 0
    centre = origin + dims * cs / 2
    pos = centre + rng.uniform(-1, 1, 3) * dims * cs * 0.7
    if seed % 2: pos = np.round(pos)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    if seed % 7 == 0: q = np.array([0.0, 0.0, 0.0, 1.0])
    lens = st["fov"] * np.pi / 8
    cam = camera_for(sc, settings_store(st), pos, q, lens)
    try:
        r = cam.render(0, want_rays=True)
    except nat.VrtError as e:
        return 'vrterror: ' + str(e)[:80]
    o = ol.render(sc, st, pos, q, lens, r.pixels, libm=ol.LIBM_PORTABLE)
    got, exp = active(r), o["rays"]
    assert len(got) == len(exp), 'nrays'
    for f in ("x", "y", "s", "color", "alpha", "counters", "ntrav", "detail", "energy", "step", "life", "bounces", "pos", "vel"):
        assert np.array_equal(got[f], exp[f]), (f, np.flatnonzero((got[f] != exp[f]).reshape(len(got), -1).any(1))[:5])
    assert np.array_equal(r.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32)), 'f32'
    assert np.array_equal(np.array(r.traversed(cs), np.int64).reshape(-1, 3), o["traversed"]), 'traversed'
    assert (r.stats[:8] == o["counters"]).all(), 'counters'
    # the fast kernel a frame normally uses (no ray records; resolution mode picked from the scene): per-sample
    # results, image, traversed list and counters
    if os.environ.get("SOAK_UNCACHED"):
        cam.cache_draws = False
    rf = cam.render(0, want_ray_rgba=True)
    packed = (exp["color"][:, 0] | (exp["color"][:, 1] << 8) | (exp["color"][:, 2] << 16) | (exp["alpha"] << 24)).astype(np.uint32)
    rr = rf.ray_rgba.cpu().numpy().view(np.uint32)
    slot = got_slots(r)
    assert np.array_equal(rr[slot], packed), 'fast ray_rgba'
    assert rr.sum() == packed.sum(dtype=np.uint64) or np.count_nonzero(rr) <= len(packed), 'fast unused slots'
    assert np.array_equal(rf.rgba_f32.cpu().numpy(), o["pix_mean"].astype(np.float32)), 'fast f32'
    assert np.array_equal(np.array(rf.traversed(cs), np.int64).reshape(-1, 3), o["traversed"]), 'fast traversed'
    assert (rf.stats[:9] == r.stats[:9]).all(), 'fast counters'
    # explicit-ray entry point on the same rays: Camera.trace_many with each ray's own draw stream
    sel = np.arange(len(got))[:: max(1, len(got) // 200)]
    W, H = st["width"], st["height"]
    dx = [-1 + (int(got["x"][i]) / W) * 2 for i in sel]
    dy = [-1 + (int(got["y"][i]) / H) * 2 for i in sel]
    nd = max(8, int(got["counters"][sel, 5].max()) + 1)
    draws = np.stack([ol.rng_draws((1 + int(got["x"][i])) * (1 + int(got["y"][i])) * (1 + int(got["s"][i])), nd + 1)[1:] for i in sel])
    rays = cam.trace_many(dx, dy, [float(got["detail"][i]) for i in sel], draws=draws)
    rec = cam.last_trace_records
    for f in ("color", "energy", "step", "life", "bounces", "pos", "vel"):
        assert np.array_equal(rec[f], got[f][sel]), ('trace_many', f)
    ce, ct = rec["counters"].copy(), got["counters"][sel].copy()
    ct[:, 5] -= 1                     # tile() itself consumed the lod_random draw
    assert np.array_equal(ce, ct), ('trace_many', 'counters')
    return None

bad = 0; skipped = 0; t0 = time.time()
a, b = int(sys.argv[1]), int(sys.argv[2])
for seed in range(a, b):
    try:
        m = one(seed)
        if m: skipped += 1; print('seed', seed, m, flush=True)
    except Exception as e:
        bad += 1; print('seed', seed, 'FAILED', type(e).__name__, str(e)[:300], flush=True)
    if seed % 100 == 0: print('seed', seed, 'elapsed %.0f s' % (time.time() - t0), flush=True)
print('done, failures:', bad, 'skipped', skipped)
