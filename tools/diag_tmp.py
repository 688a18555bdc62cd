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
r = cam.render(0, want_traversed=True)
s = r.stats
rays=s[8]; waves = 16*4096
print('rays',rays,'inner iters/wave %.1f'%(s[12]/waves),'avg marching lanes %.1f'%(s[13]/s[12]),'outer iters/wave %.1f'%(s[14]/waves),
      'hit lanes per outer %.1f'%((s[15]>>16)/s[14]), 'ended lanes per outer %.1f'%((s[15]&0xffff)/s[14]) )
print('steps/ray %.1f hits/ray %.2f'%(s[6]/rays, s[4]/rays))
