"""Ad-hoc: time rank 0's share of a G-way (x ^ y) % G partition vs a row-interleaved and a block partition (1 GPU)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from python_raytracer_amd import Camera
from python_raytracer_amd.data import make_settings
from python_raytracer_amd.lib import vec3, quaternion
from python_raytracer_amd.multigpu import rank_pixels
st = make_settings(width=3840, height=2160, samples=8, max_bounces=8.0, threads=1)
cam = Camera(settings=st)
scene, cam_pos, cam_rot, mats = bench.load_default_scene()
cam.set_packed_scene(scene); cam.pos, cam.rot = vec3(*cam_pos.tolist()), quaternion(*cam_rot.tolist())
def timeit(px, label):
    dp = cam.upload_pixels(px)
    for _ in range(2): cam.render(0, pixels=dp, check=False, want_traversed=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): r = cam.render(0, pixels=dp, check=False, want_traversed=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print('%-28s %8d px  %.2f ms  -> x8 = %.2f ms' % (label, len(px), dt*1e3, dt*8e3))
full = rank_pixels(3840, 2160, 1, 0)
timeit(full, 'full frame')
timeit(rank_pixels(3840, 2160, 8, 0), '(x^y)%8 == 0')
timeit(full[(full[:,1] % 8) == 0], 'y%8 == 0 (row interleave)')
timeit(full[(full[:,0] % 8) == 0], 'x%8 == 0 (col interleave)')
timeit(full[full[:,0] < 480], 'x < 480 (block)')
