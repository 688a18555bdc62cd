#!/usr/bin/env python3
"""bench.py -- the per-pixel voxel trace hot path on MI355X (BASELINE.json metric: Mrays/s primary+bounce).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5] [--no-cpu] [--reseed] [--no-context]

A step = one frame: every rank renders its pixel shard (seed classes, or (x ^ y) % N == rank, the reference's own
partition, with --partition xor) of the SAME frame through Camera.render -> vrt_render_tile (HIP), then the ranks'
RGBA8 tiles (as int32 words) are gathered to rank 0 (RCCL) and scattered to image order.  Inputs (scene, pixel
lists, and -- static seeds being the reference default -- the frame-invariant draw and ray tables) are resident in
HBM before the timed region; --reseed rebuilds both tables inside every timed frame.  Prints ONE JSON line on rank 0.

Default workload = BASELINE config 3 (mods/default scene fixture, 3840x2160, samples 8, 8 bounces): the
configuration the north star's Mrays/s targets are quoted on; c2 / c5 are selectable.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (width, height, samples, max_bounces, scene, overrides)
    "c1": dict(width=96, height=54, samples=1, max_bounces=2, scene="default",
               label="mods/default 96x54 spp1 max_bounces2 (BASELINE config 1)"),
    "c2": dict(width=1920, height=1080, samples=1, max_bounces=4, scene="default",
               label="mods/default 1920x1080 spp1 max_bounces4 (BASELINE config 2)"),
    "c3": dict(width=3840, height=2160, samples=8, max_bounces=8, scene="default",
               label="mods/default 3840x2160 spp8 max_bounces8 (BASELINE config 3/4)"),
    # not BASELINE configurations: launch sizes in between, to check the scheduling defaults away from the sizes they were
    # tuned on (tools/policy_check.py)
    "x1": dict(width=2560, height=1440, samples=4, max_bounces=8, scene="default",
               label="mods/default 2560x1440 spp4 max_bounces8 (policy check, not a BASELINE config)"),
    "x2": dict(width=1920, height=1080, samples=16, max_bounces=6, scene="default",
               label="mods/default 1920x1080 spp16 max_bounces6 (policy check, not a BASELINE config)"),
    "x3": dict(width=1280, height=720, samples=8, max_bounces=8, scene="default",
               label="mods/default 1280x720 spp8 max_bounces8 (policy check, not a BASELINE config)"),
    "c5": dict(width=4096, height=4096, samples=16, max_bounces=8, scene="synth1024",
               label="synthetic 1024^3 dense volume 4096x4096 spp16 max_bounces8 (BASELINE config 5)",
               over=dict(dist_max=1024, dist_min=0, dof=0.0, lod_edge=0.0, lod_random=0.0, lod_samples=0.0,
                         lod_bounces=0.0)),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def load_default_scene(world=False):
    """The flattened mods/default scene.  world=False: the camera's chunks as the reference's Window.chunk_update selected
    them for the default camera (per-chunk resolutions 1 and 2 in the fixture).  world=True: every world chunk at full
    resolution, for Camera.set_world_scene + chunk_update (the selection then happens on the device, per frame)."""
    from python_raytracer_amd import PackedScene
    z = np.load(os.path.join(ROOT, "tests", "golden", "scene_default.npz"))
    res = np.where(z["present"] != 0, 1, 0).astype(z["res"].dtype) if world else z["res"]
    sc = PackedScene.from_dense(z["origin"], z["dims"], int(z["chunk_size"][0]), z["present"], res,
                                z["grid_lod0"], z["materials"])
    return sc, z["cam_pos"], z["cam_rot"], z["materials"]


def make_synth_scene(n, materials, device):
    """1024^3 (or n^3) hashed volume generated on the device straight into the packed layout."""
    import torch
    from python_raytracer_amd import PackedScene, _native as nat
    cs = 16
    d = n // cs
    table = torch.zeros(d * d * d, dtype=torch.int32, device=device)
    vox = torch.zeros(n * n * n, dtype=torch.uint8, device=device)
    nat.check(nat.lib().vrt_synth_volume(n, cs, table.data_ptr(), vox.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "vrt_synth_volume")
    return PackedScene.from_device([-n // 2] * 3, [d] * 3, cs, table, vox, d * d * d, materials, max_resolution=1)


def single_gpu_reference(config):
    """The newest committed single-GPU bench line of this configuration (profiles/rNN_vM_bench_<config>.json, written
    by this script on an MI355X): what a multi-GPU run of the same frame must reproduce -- frame hash and ray counts."""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_v*_bench_%s.json" % config)):
        m = re.match(r"r(\d+)_v(\d+)_bench_", os.path.basename(path))
        if not m:
            continue
        try:
            d = json.load(open(path))
        except ValueError:
            continue
        if d.get("n_gpus") == 1 and d.get("config", {}).get("image_sha256"):
            key = (int(m.group(1)), int(m.group(2)))
            if best is None or key > best[0]:
                best = (key, os.path.basename(path), d)
    return best


def b_alg(stats, n_px):
    """Algorithmic bytes: 1 B per voxel read, 8 B per chunk-table read, 32 B per material record,
    8 B per random draw, 20 B per pixel written (fp32 RGBA + RGBA8)."""
    lookup, nbr, resnap, chunk_get, hit, draw = (int(stats[i]) for i in range(6))
    return (lookup + nbr) + 8 * (resnap + chunk_get) + 32 * hit + 8 * draw + 20 * n_px


def host_volume(scene, materials):
    """The synthetic volume as the oracle wants it (a dense [x][y][z] id grid), copied back from the device and taken out
    of the packed brick order (vrt.h: blocks [cx][cy][cz] of 8^3 bricks of 4^3 micro-bricks).  The device generator is
    checked byte for byte against oracle_lib.synth_scene's numpy one by tests/test_gpu_parity.py; generating 1024^3 ids
    with numpy would take two minutes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    cs = int(scene.chunk_size)
    d, nb = int(scene.dims[0]), cs // 8
    g = scene.device_tensors["voxels"].cpu().numpy().reshape(d, d, d, nb, nb, nb, 2, 2, 2, 4, 4, 4)
    g = np.ascontiguousarray(g.transpose(0, 3, 6, 9, 1, 4, 7, 10, 2, 5, 8, 11)).reshape(d * cs, d * cs, d * cs)
    return ol.Scene([int(v) for v in scene.origin], [d] * 3, cs, np.ones((d, d, d), np.uint8), np.ones((d, d, d), np.uint8), g,
                    np.asarray(materials, np.float64))


def cpu_baseline(cfg, st_dict, cam_pos, cam_rot, lens, stride, host_scene=None):
    """The CPU oracle (C restatement of the reference path, glibc libm = the reference's arithmetic) on every
    `stride`-th pixel in x and y of the same frame (~10-30 CPU-seconds), up to 16 host threads, pixels dealt
    round-robin like the reference's settings.pixels.  host_scene: the oracle's scene (default: the mods/default fixture)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    sc = host_scene if host_scene is not None else ol.default_scene()
    threads = min(len(os.sched_getaffinity(0)), 16)   # one GPU's share of the host (16 cores per GPU on the box)
    xs, ys = np.meshgrid(np.arange(0, cfg["width"], stride), np.arange(0, cfg["height"], stride), indexing="ij")
    sub = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    t0 = time.time()
    o = ol.render(sc, st_dict, cam_pos, cam_rot, lens, sub, libm=ol.LIBM_GLIBC, threads=threads, want_rays=False,
                  want_traversed=False)
    dt = time.time() - t0
    c = o["counters"]
    rays = int(o["n_rays"]) + int(c[4]) - int(c[7])
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample="every %d-th pixel in x and y of the same frame: %d pixels, %d primary rays, %.2f s wall"
                       % (stride, len(sub), int(o["n_rays"]), dt),
                primary_rays_per_s=int(o["n_rays"]) / dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--partition", default="seed", choices=["seed", "xor", "tiles"],
                    help="pixel partition across ranks for N > 1 (multigpu.owner_map)")
    ap.add_argument("--reseed", action="store_true",
                    help="re-seed MT19937 and regenerate the ray table inside every timed frame (what a non-static run "
                         "must do) instead of building the static-seed tables once (vrt_draw_table_build, "
                         "vrt_ray_table_build; the default, like the reference's static = true)")
    ap.add_argument("--rng-cache", action="store_true", help="accepted for compatibility: it is the default now")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2, 3],
                    help="frames submitted on that many HIP streams in turn: with 2, frame k + 1 starts while frame k's "
                         "last waves drain (throughput mode; the per-kernel durations then overlap)")
    ap.add_argument("--no-context", action="store_true",
                    help="skip the re-seeded context frames after the timed region (profiling runs: every launch a "
                         "profiler sees is then a warm-up or a timed frame)")
    ap.add_argument("--no-traversed", action="store_true", help="do not record traversed chunks")
    ap.add_argument("--world-flow", action="store_true",
                    help="default scene only: keep the whole world resident (Camera.set_world_scene) and let every frame's "
                         "Camera.chunk_update select the camera's chunks and their LOD on the device -- the reference's "
                         "own flow (Window.chunk_update, init.py:441-452) -- instead of rendering the pre-selected fixture; "
                         "the frame must come out identical")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from python_raytracer_amd import Camera, _native as nat
    from python_raytracer_amd.data import make_settings
    from python_raytracer_amd.lib import vec3, quaternion
    from python_raytracer_amd.multigpu import rank_pixels, TileGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
    # one process per GPU; VRT_BENCH_BACKEND=gloo (ranks may then share a GPU) exists only to rehearse the N > 1 code
    # path on a one-GPU box
    backend = os.environ.get("VRT_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cfg = CONFIGS[args.config]
    over = dict(cfg.get("over", {}))
    st = make_settings(width=cfg["width"], height=cfg["height"], samples=cfg["samples"],
                       max_bounces=float(cfg["max_bounces"]), threads=1, **over)
    st.pixels = None  # the rank's pixel list is passed explicitly
    cam = Camera(settings=st, device=local_rank)
    cam.cache_draws = not args.reseed
    if cfg["scene"] == "default":
        scene, cam_pos, cam_rot, mats = load_default_scene(world=args.world_flow)
        cam.pos, cam.rot = vec3(*cam_pos.tolist()), quaternion(*cam_rot.tolist())
        if args.world_flow:
            st.culling = False          # (the fixture was selected with culling off)
            cam.set_world_scene(scene)
            cam.chunk_update(None)
        else:
            cam.set_packed_scene(scene)
    else:
        _, _, _, mats = load_default_scene()
        cam.set_packed_scene(make_synth_scene(1024, mats, dev))
        cam.pos, cam.rot = vec3(0.5, 0.5, 0.5), quaternion(0.0, 0.0, 0.0, 1.0)
        cam_pos, cam_rot = np.array([0.5, 0.5, 0.5]), np.array([0.0, 0, 0, 1])

    # rank's pixels: seed-class partition by default for N > 1 (python_raytracer_amd/multigpu.py), the reference's
    # (x ^ y) % N with --partition xor; the tiles travel as RGBA8 like the reference's tile bytes (init.py:185-190)
    partition = args.partition if world > 1 else "xor"
    pixels = rank_pixels(st.width, st.height, world, rank, partition, st.samples)
    gather = TileGather(st.width, st.height, 1, torch.int32, dev, partition=partition, samples=st.samples) \
        if world > 1 else None
    L = nat.lib()
    last = {}
    pixels_dev = cam.upload_pixels(pixels)  # resident in HBM before the timed region

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.frames_in_flight)] if args.frames_in_flight > 1 else []
    turn = [0]

    def step():
        if args.world_flow:  # the reference re-selects the camera's chunks every chunk_rate ms; here: every frame
            cam.chunk_update(last.get("r") if st.culling else None)
        if streams:  # frame k on stream k % F: its march overlaps the previous frame's draining waves
            cur = streams[turn[0] % len(streams)]
            turn[0] += 1
            with torch.cuda.stream(cur):
                r = cam.render(0, pixels=pixels_dev, want_image=True, want_f32=True,
                               want_traversed=not args.no_traversed, check=False)
                if gather is not None:
                    if gather.pending is not None:
                        last["image"] = gather.collect()
                    gather.submit(window=r.image_u8.view(torch.int32))
            last["r"] = r
            return
        r = cam.render(0, pixels=pixels_dev, want_image=True, want_f32=True, want_traversed=not args.no_traversed,
                       check=False)
        if gather is not None:  # frame k's gather overlaps frame k + 1's render
            if gather.pending is not None:
                last["image"] = gather.collect()
            gather.submit(window=r.image_u8.view(torch.int32))
        last["r"] = r

    def drain():
        if gather is not None and gather.pending is not None:
            last["image"] = gather.collect()

    # one checked frame first: validates the run and lets the camera pick its draw-table width (32 | 64)
    cam.render(0, pixels=pixels_dev, want_image=False, want_f32=False, want_traversed=False, check=True)
    torch.cuda.synchronize()  # (the tables it built are read from the other streams)
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events around the march launches only (the roofline's kernel): an event pair costs a frame ~12 us, and the three
    # pairs of a config-2 frame (march, re-trace tiers, resolve) were 6 % of it.  The other kernels are timed in frames of
    # their own after the timed region (below).  VRT_BENCH_EVENTS=0: none at all (the roofline is then empty).
    if os.environ.get("VRT_BENCH_EVENTS", "1") != "0":
        L.vrt_profile_begin_kinds(1 << 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = (C.c_double * nat.NPROF)()
    launches = (C.c_int64 * nat.NPROF)()
    L.vrt_profile_end(ms, launches)
    # every kernel kind, in up to 5 more frames of the same kind outside the timed region
    other_frames = min(args.steps, 5)
    ms_all = (C.c_double * nat.NPROF)()
    launches_all = (C.c_int64 * nat.NPROF)()
    L.vrt_profile_begin()
    for _ in range(other_frames):
        step()
    drain()
    torch.cuda.synchronize()
    L.vrt_profile_end(ms_all, launches_all)

    stats = last["r"]._stats_dev.cpu().numpy()
    # SHA-256 of the finished RGBA8 frame (rank 0): identical for every N and partition, or the run is wrong
    import hashlib
    frame = last.get("image") if world > 1 else last["r"].image_u8
    image_sha = hashlib.sha256(frame.contiguous().cpu().numpy().tobytes()).hexdigest() if frame is not None else None
    if stats[nat.S_RNG_EXHAUSTED]:
        raise SystemExit("invalid run: %d rays exhausted the random-draw tables" % stats[nat.S_RNG_EXHAUSTED])
    if stats[nat.S_TRAV_OUTSIDE] or stats[nat.S_STALLED]:
        raise SystemExit("invalid run: march reported internal errors (outside the traversed box, stalled waves) %r"
                         % stats[[nat.S_TRAV_OUTSIDE, nat.S_STALLED]].tolist())
    # rays = primary + bounce (shader invocations after which the march continued), SURVEY.md 8d
    local = np.array([int(stats[8]), int(stats[4]) - int(stats[7]), dt] + [int(v) for v in stats[:8]], np.float64)
    if world > 1:
        t = torch.from_numpy(local).to(dev)
        mx = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        tot, dt = t.cpu().numpy(), float(mx[2])
    else:
        tot = local
    if rank != 0:
        dist.destroy_process_group()
        return
    primary, bounce = int(tot[0]), int(tot[1])
    per_step = dt / args.steps
    value = (primary + bounce) / per_step / 1e6

    # roofline of the dominant kernel (march_kernel) on rank 0: algorithmic bytes per launch / mean launch time
    n_march = int(launches[1]) or 1
    march_ms = ms[1] / n_march
    balg_frame = b_alg(stats, len(pixels))
    balg_launch = balg_frame * args.steps / n_march
    achieved = balg_launch / (march_ms * 1e-3) / 1e9 if march_ms > 0 else 0.0
    # HBM traffic of one march launch from the rocprofv3 counters of a SEPARATE run of this command (profiles/
    # pmc_<config>.json, tools/save_profiles.py): not measured in this process
    traffic = traffic_hi = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_%s.json" % args.config)
    if os.path.exists(pmc_path) and not args.reseed:
        pmc = json.load(open(pmc_path))
        traffic, traffic_hi = pmc.get("hbm_bytes_per_march_launch"), pmc.get("hbm_bytes_per_march_launch_upper")
    # kernel variant of the frame march: resolution mode (0: all chunks at resolution 1, 1: <= 2, 2: any) and speculation
    # depth (8 steps for modes 0 and 1 and for voxel data far beyond the caches, else 4: march_deep() in vrt_kernels.hip)
    sc_ = cam._ensure_scene()
    deep_env = os.environ.get("VRT_SPEC_DEEP")
    res_mode_ = {1: 0, 2: 1}.get(int(cam._c_scene(sc_).max_resolution), 2)   # (what vrt_render_tile was told)
    deep = (int(deep_env) != 0) if deep_env is not None else (res_mode_ != 2 or sc_.n_slots * int(st.chunk_size) ** 3 > (512 << 20))
    spec_depth = 8 if deep else 4
    res_mode = res_mode_
    out = {
        "metric": "Mrays/s (primary+bounce)", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per_step * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "mods/default scene fixture (tests/golden/scene_default.npz), static seeds"
                if cfg["scene"] == "default" else "synthetic hashed 1024^3 volume generated on device",
        "config": {"workload": cfg["label"], "width": st.width, "height": st.height, "samples": st.samples,
                   "max_bounces": st.max_bounces, "primary_rays": primary, "bounce_rays": bounce,
                   "primary_Mrays_per_s": round(primary / per_step / 1e6, 3),
                   "partition": {"xor": "(x ^ y) %% %d" % world, "seed": "seed classes over %d ranks" % world,
                                 "tiles": "8x8 pixel blocks over %d ranks" % world}[partition], "traversed": not args.no_traversed,
                   "fast_draws": cam.fast_draws, "image_sha256": image_sha, "frames_in_flight": args.frames_in_flight,
                   "scene_flow": "world resident, Camera.chunk_update selects chunks + LOD on the device every frame"
                                 if args.world_flow else "pre-selected camera chunks (fixture)",
                   "rng_retraced_rays": int(stats[nat.S_RNG_RETRACED]),
                   "rng_table": "re-seeded and ray table regenerated in every timed frame (--reseed)" if args.reseed else
                                "static seeds: draw table + ray table built once before the timed region, reused"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                     "traffic_source": "profiles/pmc_%s.json: rocprofv3 FETCH_SIZE x 1 (calibrated for this kernel's 1/8-byte "
                                       "gathers, profiles/r02_fetch_size_calibration.json) + WRITE_SIZE, separate run" % args.config
                     if traffic else None,
                     "kernel": ("march_pool_kernel<%d,%d>" if stats[nat.S_POOL_GROUPS] else "march_kernel<%d,%d,false,false,0>")
                               % (spec_depth, res_mode), "launches": n_march,
                     "avg_launch_ms": round(march_ms, 4), "alg_bytes_per_launch": int(balg_launch),
                     # counter bytes over this run's launch time; the upper bound doubles FETCH_SIZE (the guide's factor
                     # for wide streaming reads, which this kernel's ray-table and draw reads partly are)
                     "counter_hbm_GBps": [round(t / (march_ms * 1e-3) / 1e9, 1) if t and march_ms > 0 else None
                                          for t in (traffic, traffic_hi)],
                     "alg_bytes_per_primary_ray": round(balg_frame / max(1, int(stats[8])), 2)},
        # march: HIP events inside the timed region; the others: the same frames again after it, every kernel bracketed
        "kernel_ms_per_step": {nat.PROF_NAMES[k]: round(ms[k] / args.steps if k == 1 else ms_all[k] / other_frames, 4)
                               for k in range(len(nat.PROF_NAMES))},
    }
    exit_code = 0
    if world > 1:
        # self-check of a multi-GPU run: the gathered frame and the whole-job ray counts must be those of the committed
        # single-GPU line of this configuration (the same frame: static seeds, the same scene and camera)
        ref = single_gpu_reference(args.config)
        if ref is None:
            out["matches_single_gpu"] = None
        else:
            rc_ = ref[2]["config"]
            same = (rc_["image_sha256"] == image_sha and int(rc_["primary_rays"]) == primary and
                    int(rc_["bounce_rays"]) == bounce)
            out["matches_single_gpu"] = bool(same)
            out["single_gpu_reference"] = {"line": "profiles/" + ref[1], "image_sha256": rc_["image_sha256"],
                                           "primary_rays": int(rc_["primary_rays"]), "bounce_rays": int(rc_["bounce_rays"]),
                                           "value": ref[2]["value"]}
            if not same:
                exit_code = 3
    if world == 1 and not args.reseed and st.static and not args.no_context:
        # context, outside the timed region above: the same K frames with both tables rebuilt in every frame
        cam.cache_draws = False
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        per2 = (time.perf_counter() - t1) / args.steps
        cam.cache_draws = True
        out["reseeded_every_frame"] = {"ms_per_step": round(per2 * 1e3, 4),
                                       "value": round((primary + bounce) / per2 / 1e6, 3), "unit": "Mrays/s"}
    if world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        st_dict = ol.make_settings(width=st.width, height=st.height, samples=st.samples,
                                   max_bounces=float(st.max_bounces), **over)
        # sample sized for ~10-30 s of CPU work (every 16th pixel of the synthetic volume's frame)
        stride = 16 if cfg["scene"] != "default" else (4 if st.width * st.height * st.samples > 8_000_000 else 1)
        out["cpu_baseline"] = cpu_baseline(cfg, st_dict, cam_pos, cam_rot, cam.lens, stride,
                                           None if cfg["scene"] == "default" else host_volume(cam._ensure_scene(), mats))
        out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 4)
        ref = os.path.join(ROOT, "tests", "golden", "ref_timing.json")
        if os.path.exists(ref):  # the genuine Python reference, measured in the build container (context only)
            rt = json.load(open(ref))
            out["cpu_baseline"]["python_reference_primary_rays_per_s_8_cores_config1"] = round(rt["pool8_primary_rays_per_s"], 1)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if exit_code:
        sys.stderr.write("bench.py: the %d-GPU frame differs from the committed single-GPU line (%s)\n"
                         % (world, out["single_gpu_reference"]["line"]))
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
