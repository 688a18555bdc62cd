"""Soak (GPU box): Camera.chunk_update (vrt_select_chunks) on random worlds, cameras and traversed lists against the
oracle.  usage: soak_select.py FIRST_SEED LAST_SEED"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle_lib as ol
from gpu_util import camera_for, settings_store
from python_raytracer_amd import _native as nat
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    try:
        # ---- chunk selection
        cs = int(rng.choice([8, 16, 32]))
        dims = rng.integers(1, 7, 3)
        origin = (rng.integers(-4, 3, 3) * cs).astype(np.int64)
        present = (rng.random(tuple(dims)) < 0.7).astype(np.uint8)
        pos = origin + rng.uniform(-0.5, 1.5, 3) * dims * cs
        dist_max = float(rng.choice([16, 48, 192, 1000]))
        lod = int(rng.choice([0, 1, 2, 5]))
        culling = bool(rng.integers(0, 2))
        ntr = int(rng.integers(0, 40))
        trav = (origin + rng.integers(-1, dims.max() + 1, (ntr, 3)) * cs).astype(np.float64)
        p0, r0 = ol.select_chunks(origin, dims, cs, present, pos, dist_max, lod, culling, trav)
        from python_raytracer_amd import Camera, PackedScene
        from python_raytracer_amd.lib import vec3, quaternion
        st = ol.make_settings(width=8, height=8, chunk_size=cs, dist_max=dist_max, chunk_lod=lod)
        sst = settings_store(st); sst.culling = culling
        cam = Camera(settings=sst)
        grid = np.zeros(tuple(dims * cs), np.uint8)
        mats = np.array([[1, 2, 3, 0, 1, 1, 0.0]])
        blocks_present = present.copy()
        for c in np.argwhere(present): grid[tuple(c * cs)] = 1
        world = PackedScene.from_dense(origin, dims, cs, present, np.ones_like(present), grid, mats)
        cam.set_world_scene(world)
        cam.pos = vec3(*[float(v) for v in pos]); cam.rot = quaternion(0.0, 0.0, 0.0, 1.0)
        table = cam.chunk_update([tuple(int(v) for v in t) for t in trav]).cpu().numpy().view(np.uint32).reshape(tuple(dims))
        assert np.array_equal((table != 0).astype(np.uint8), p0), 'select present'
        assert np.array_equal((table >> 24).astype(np.uint8)[table != 0], r0[p0 > 0]), 'select res'
    except Exception as e:
        bad += 1; print('seed', seed, 'FAILED', type(e).__name__, str(e)[:200], flush=True)
print('done, failures:', bad)
