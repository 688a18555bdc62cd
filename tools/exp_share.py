"""Ad-hoc: per-kernel time of rank 0's 1/8 share of config 3 (one GPU), to see the fixed per-frame costs."""
import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import bench
from python_raytracer_amd import Camera, _native as nat
from python_raytracer_amd.data import make_settings
from python_raytracer_amd.lib import vec3, quaternion
from python_raytracer_amd.multigpu import rank_pixels
st = make_settings(width=3840, height=2160, samples=8, max_bounces=8.0, threads=1)
cam = Camera(settings=st)
scene, cam_pos, cam_rot, mats = bench.load_default_scene()
cam.set_packed_scene(scene); cam.pos, cam.rot = vec3(*cam_pos.tolist()), quaternion(*cam_rot.tolist())
L = nat.lib()
import os
for world, part in [(int(w), pt) for w in os.environ.get("EXP_WORLDS", "32,16,8,4,2,1").split(",")
                    for pt in os.environ.get("EXP_PARTS", "seed").split(",")]:
    px = rank_pixels(3840, 2160, world, 0, part, 8)
    dp = cam.upload_pixels(px)
    for trav in (True,):
        fif = int(os.environ.get("EXP_FIF", "1"))
        streams = [torch.cuda.Stream() for _ in range(fif)]
        def frame(i):
            if fif == 1:
                return cam.render(0, pixels=dp, check=False, want_traversed=trav)
            with torch.cuda.stream(streams[i % fif]):
                return cam.render(0, pixels=dp, check=False, want_traversed=trav)
        cam.render(0, pixels=dp, check=False, want_traversed=trav); torch.cuda.synchronize()
        for i in range(3): frame(i)
        # like bench.py: the timed frames carry HIP events around the march only (EXP_EVENTS=all: around every kernel, as
        # before round 4's last set; none: no events); the other kernels' times come from frames of their own afterwards
        ev = os.environ.get("EXP_EVENTS", "march")
        torch.cuda.synchronize()
        if ev != "none": L.vrt_profile_begin_kinds(0xffffffff if ev == "all" else 2)
        t = time.perf_counter()
        n = 30
        for i in range(n): r = frame(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        ms = (C.c_double * nat.NPROF)(); la = (C.c_int64 * nat.NPROF)(); L.vrt_profile_end(ms, la)
        if ev == "march":
            m2 = (C.c_double * nat.NPROF)(); L.vrt_profile_begin()
            for i in range(n): frame(i)
            torch.cuda.synchronize(); L.vrt_profile_end(m2, la)
            for k in range(nat.NPROF):
                if k != 1: ms[k] = m2[k]
        print('world %d %s traversed %d: %.3f ms/frame; kernels %s sum %.3f' % (world, part, trav, dt * 1e3,
              {nat.PROF_NAMES[k]: round(ms[k] / n, 3) for k in range(5)}, sum(ms) / n))
