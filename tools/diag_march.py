"""Loop statistics of march_kernel (needs a build with -DVRT_DIAG, see vrt_kernels.hip): passes per wave, lanes
marching per pass, lanes shading per slow-body execution and rough cycle shares.  Diagnostic only."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bench, torch
from python_raytracer_amd import Camera
from python_raytracer_amd.data import make_settings
from python_raytracer_amd.lib import vec3, quaternion
st = make_settings(width=3840,height=2160,samples=8,max_bounces=8.0,threads=1); 
cam = Camera(settings=st)
scene, cam_pos, cam_rot, mats = bench.load_default_scene()
cam.set_packed_scene(scene); cam.pos, cam.rot = vec3(*cam_pos.tolist()), quaternion(*cam_rot.tolist())
r = cam.render(0, want_traversed=True, check=False)
s = r._stats_dev.cpu().numpy().astype(np.uint64)
waves = 4096  # one launch per frame x waves per launch
rays = int(s[8]) & 0xffffffff
cyc = [int(s[9]), int(s[10]), int(s[11]), int(s[8])>>32]
tot = sum(cyc)
print('inner iters/wave %.1f'%(int(s[12])/waves),'avg marching lanes %.1f'%(int(s[13])/int(s[12])),'outer iters/wave %.1f'%(int(s[14])/waves),
      'hit lanes per outer %.1f'%(int(s[15])/int(s[14])))
print('cycle shares: refill %.1f%% march %.1f%% hit %.1f%% ended %.1f%%; cycles/wave %.0f'%(tuple(100*c/tot for c in cyc)+(tot/waves,)))
print('per inner iter cycles %.0f; per outer: refill %.0f hit %.0f ended %.0f'%(cyc[1]/int(s[12]), cyc[0]/int(s[14]), cyc[2]/int(s[14]), cyc[3]/int(s[14])))
