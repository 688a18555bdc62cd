"""How many re-snaps still read their traversed cell's key from memory, and how many visits fall outside the settled bitmap's
window.  Needs the counting build: tools/build_variant.sh cnt -DVRT_COUNT_UNSETTLED, then on the GPU box
    VRT_SO=$PWD/python_raytracer_amd/_vrt_cnt.so python tools/count_key_reads.py c5|c3
(that build adds the two counts into stats[14] and stats[15]; diagnostic only)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, bench
from python_raytracer_amd import Camera
from python_raytracer_amd.data import make_settings
from python_raytracer_amd.lib import vec3, quaternion
cfg = bench.CONFIGS[sys.argv[1]]
st = make_settings(width=cfg["width"], height=cfg["height"], samples=cfg["samples"], max_bounces=float(cfg["max_bounces"]), threads=1, **cfg.get("over", {}))
cam = Camera(settings=st)
scene, cam_pos, cam_rot, mats = bench.load_default_scene()
if cfg["scene"] == "default":
    cam.set_packed_scene(scene); cam.pos, cam.rot = vec3(*cam_pos.tolist()), quaternion(*cam_rot.tolist())
else:
    cam.set_packed_scene(bench.make_synth_scene(1024, mats, torch.device("cuda", 0))); cam.pos, cam.rot = vec3(0.5, 0.5, 0.5), quaternion(0.0, 0.0, 0.0, 1.0)
cam.render(0, check=True)
r = cam.render(0, check=False)
s = r._stats_dev.cpu().numpy()
print(sys.argv[1], "visits", int(s[2]), "key reads", int(s[14]), "visits outside the window", int(s[15]), "box", r.trav_dims)
