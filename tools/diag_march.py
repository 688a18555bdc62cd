"""Loop statistics of march_kernel from the instrumented build (VRT_DIAG=1 -> python_raytracer_amd/_vrt_diag.so,
-DVRT_DIAG): passes per wave, lanes active per body, cycle shares per phase.  Diagnostic only; usage on the GPU box:
    VRT_DIAG=1 python tools/diag_march.py [c3|c5|c2]
DIAG_RESEED=1: the uncached frame (Camera.cache_draws = False: no ray table, the march's refill makes the records)."""
import ctypes as C
import os
import sys

os.environ.setdefault("VRT_DIAG", "1")  # VRT_DIAG=2: with the speculation histograms
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from python_raytracer_amd import Camera, _native as nat
from python_raytracer_amd.data import make_settings
from python_raytracer_amd.lib import vec3, quaternion

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = bench.CONFIGS[cfgname]
st = make_settings(width=cfg["width"], height=cfg["height"], samples=cfg["samples"], max_bounces=float(cfg["max_bounces"]),
                   threads=1, **cfg.get("over", {}))
cam = Camera(settings=st)
cam.cache_draws = os.environ.get("DIAG_RESEED", "0") == "0"
scene, cam_pos, cam_rot, mats = bench.load_default_scene()
if cfg["scene"] == "default":
    cam.set_packed_scene(scene)
    cam.pos, cam.rot = vec3(*cam_pos.tolist()), quaternion(*cam_rot.tolist())
else:
    cam.set_packed_scene(bench.make_synth_scene(1024, mats, torch.device("cuda", 0)))
    cam.pos, cam.rot = vec3(0.5, 0.5, 0.5), quaternion(0.0, 0.0, 0.0, 1.0)
L = nat.lib()
L.vrt_diag_read.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 96)()
cam.render(0, want_traversed=True, check=True)     # builds the tables
L.vrt_diag_read(buf, 96)
r = cam.render(0, want_traversed=True, check=True)
n = L.vrt_diag_read(buf, 96)
names = ["passes", "cyc_refill", "cyc_march", "cyc_hit", "cyc_end", "iters", "march_lanes", "hit_exec", "hit_lanes", "end_exec",
         "end_lanes", "refill_exec", "refill_lanes", "wave_cycles", "snap_iters", "snap_lanes", "brick_visits", "swaps",
         "swap_lanes", "evict_lanes", "cyc_swap", "void_lanes"] + ["nv%d" % k for k in range(1, 9)] + ["h%d" % k for k in range(1, 9)]
n_base = names.index("void_lanes") + 1
if n < len(names) + 5:  # the build without the histograms
    names = names[:n_base]
d = {k: int(buf[i]) for i, k in enumerate(names)}
rays = int(r.stats[8])
c = r.counters()
tot = d["cyc_refill"] + d["cyc_march"] + d["cyc_hit"] + d["cyc_end"]
print(cfgname, "rays", rays, "counters", c)
print("frame march: %s%s" % ("march_pool_kernel (%d workgroups%s)" % (int(r.stats[12]) & 0xffffffff, ", tiled hand-out" if int(r.stats[12]) >> 32 else "") if r.stats[12] else "march_kernel (one ray per lane)",
                             ", look-ahead across chunk borders (march_step_w)" if r.stats[14] else ""))
print("per ray: passes*64 %.2f  march iters*lanes %.2f  steps %.2f  hits %.2f" % (
    d["passes"] * 64 / rays, d["march_lanes"] / rays, (c["lookup"] + 0.0) / rays, c["hit"] / rays))
print("lanes per execution: march %.1f  hit %.1f  end %.1f  refill %.1f" % (
    d["march_lanes"] / max(1, d["iters"]), d["hit_lanes"] / max(1, d["hit_exec"]), d["end_lanes"] / max(1, d["end_exec"]),
    d["refill_lanes"] / max(1, d["refill_exec"])))
print("executions per pass: march iters %.2f  hit %.2f  end %.2f  refill %.2f" % (
    d["iters"] / d["passes"], d["hit_exec"] / d["passes"], d["end_exec"] / d["passes"], d["refill_exec"] / d["passes"]))
print("cycle shares: refill %.1f%%  march %.1f%%  hit %.1f%%  end %.1f%%   (sum/wave_cycles %.3f)" % (
    100 * d["cyc_refill"] / tot, 100 * d["cyc_march"] / tot, 100 * d["cyc_hit"] / tot, 100 * d["cyc_end"] / tot,
    tot / d["wave_cycles"]))
print("cycles per execution: march iter %.0f  hit %.0f  end %.0f  refill %.0f" % (
    d["cyc_march"] / max(1, d["iters"]), d["cyc_hit"] / max(1, d["hit_exec"]), d["cyc_end"] / max(1, d["end_exec"]),
    d["cyc_refill"] / max(1, d["refill_exec"])))
print("re-snaps: lanes per march iteration that has any %.1f (%.2f of the iterations)" % (
    d["snap_lanes"] / max(1, d["snap_iters"]), d["snap_iters"] / max(1, d["iters"])))
if d["swaps"]:
    print("ray pool: exchanges per pass %.2f, rays per exchange %.1f (of them parked into free slots %.1f), rays moved per ray %.2f, "
          "cycles per exchange %.0f (%.1f%% of the wave cycles)" % (
              d["swaps"] / d["passes"], d["swap_lanes"] / d["swaps"], d["evict_lanes"] / d["swaps"], d["swap_lanes"] / rays,
              d["cyc_swap"] / d["swaps"], 100.0 * d["cyc_swap"] / d["wave_cycles"]))
if d.get("nv1"):
    print("speculation: lanes with a present chunk %.2f of the marching lanes (the others skip void: %.2f); of those, share whose "
          "k-th position was still valid: %s; share that advanced >= k: %s" % (
              d["nv1"] / max(1, d["march_lanes"]), d["void_lanes"] / max(1, d["march_lanes"]),
              " ".join("%.2f" % (d["nv%d" % k] / d["nv1"]) for k in range(1, 9)),
              " ".join("%.2f" % (d["h%d" % k] / d["nv1"]) for k in range(1, 9))))
print("wave cycles per ray %.0f" % (d["wave_cycles"] * 64 / rays))
print("8^3 brick visits per ray %.2f (SURVEY.md 8d: B_brick = 512 B x visits = %.3f GB per frame)" % (
    d["brick_visits"] / rays, 512.0 * d["brick_visits"] / 1e9))
if n >= len(names) + 5:
    M = (1 << 64) - 1
    t0, te, t1, tsum, nw = M - int(buf[len(names)]), M - int(buf[len(names) + 1]), int(buf[len(names) + 2]), int(buf[len(names) + 3]), int(buf[len(names) + 4])
    us = lambda ticks: ticks / 100.0
    print("timeline (s_memrealtime): launch %.1f us; ray queue empty after %.1f us (%.1f %%); mean wave exit at %.1f us; "
          "waves %d; idle wave time after the queue ran dry %.1f %% of the launch" % (
              us(t1 - t0), us(te - t0), 100.0 * (te - t0) / (t1 - t0), us(tsum / nw - t0), nw,
              100.0 * (t1 - tsum / nw) / (t1 - t0)))
