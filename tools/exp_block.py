"""Ad-hoc: march time of coherent sub-blocks (x < W/k) and scattered shares of config 3 (1 GPU)."""
import sys, time, ctypes as C, numpy as np, torch, os
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
def run(px, label):
    dp = cam.upload_pixels(px)
    for _ in range(3): cam.render(0, pixels=dp, check=False, want_traversed=True)
    torch.cuda.synchronize(); L.vrt_profile_begin(); n = 10
    for _ in range(n): r = cam.render(0, pixels=dp, check=False, want_traversed=True)
    torch.cuda.synchronize()
    ms = (C.c_double * nat.NPROF)(); la = (C.c_int64 * nat.NPROF)(); L.vrt_profile_end(ms, la)
    rays = int(r._stats_dev.cpu().numpy()[8])
    print('%-34s rays %9d march %.3f ms (%d launches)  %.4f ms/Mray' % (label, rays, ms[1] / n, la[1] // n, ms[1] / n / rays * 1e6), flush=True)
full = rank_pixels(3840, 2160, 1, 0)
mode = os.environ.get("EXP_MODE", "blocks")
if mode == "blocks":
    for k in (32, 16, 8, 4, 2, 1):
        for j in (0, k // 2):
            w = 3840 // k
            run(full[(full[:, 0] >= j * w) & (full[:, 0] < (j + 1) * w)], 'block %d of %d' % (j, k))
            if k == 1: break
    for k in (32, 8, 2):
        run(full[(full[:, 1] % k) == 0], 'rows y %% %d == 0' % k)
else:
    run(rank_pixels(3840, 2160, 8, 0, "seed", 8), 'seed share 1/8')
