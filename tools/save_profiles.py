#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of the last gpurun from gpurun_out/ into profiles/ (tracked) and derive
profiles/pmc_<config>.json: HBM bytes per march launch from the FETCH_SIZE / WRITE_SIZE counters (KB -> bytes).
FETCH_SIZE is taken x 1: tools/microbench calib shows it tallies exactly 64 bytes per random 1- or 8-byte gather
(profiles/r02_fetch_size_calibration.json), which is what the march's voxel reads are; the x 2 of
MI355X_MICROARCH.md holds for wide coalesced streaming reads (part of the march's ray-table reads) and is reported
as the upper bound.  The first launch of the profiled run (cold) is left out.
usage: save_profiles.py TAG [--pmc-only]   e.g. r02_v1
--pmc-only (run by tools/profile_all.sh on the GPU box between the counter passes and the bench lines): only derive
profiles/pmc_<config>.json, so that the bench lines that follow carry exactly the traffic this script commits later from
the same counter files."""
import csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize
tag = sys.argv[1]
pmc_only = "--pmc-only" in sys.argv[2:]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
os.makedirs(P, exist_ok=True)


def is_frame_march(name):
    """the frame's march: march_pool_kernel<SPEC, RES>, or march_kernel<SPEC, RES, false, false, LK> (neither records nor re-traces)"""
    if name.startswith("void march_pool_kernel<"):
        return True
    if not name.startswith("void march_kernel<"):
        return False
    args = [a.strip() for a in name[len("void march_kernel<"):].split(">")[0].split(",")]   # SPEC, RES, RECORD, LIST, ...
    return args[2] == "false" and args[3] == "false"


for cfg in ("c3", "c5", "c2"):
    found = glob.glob(os.path.join(O, "prof_%s" % cfg, "**", "*_kernel_stats.csv"), recursive=True)
    if found and not pmc_only:  # (gpurun merges into gpurun_out/: an older run's file may still lie beside the new one)
        shutil.copy(max(found, key=os.path.getmtime), os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, cfg)))
    for name in ("bench_%s.json" % cfg, "prof_%s_bench.json" % cfg, "bench_%s_reseed.json" % cfg):
        src = os.path.join(O, name)
        if os.path.exists(src) and os.path.getsize(src) and not pmc_only:
            shutil.copy(src, os.path.join(P, "%s_%s" % (tag, name)))
    dirs = [os.path.join(O, "pmc_fetch_%s" % cfg), os.path.join(O, "pmc_write_%s" % cfg)]
    if all(os.path.isdir(d) for d in dirs):
        s = summarize(dirs, drop_first=True)
        # (the march instance of the timed frames: the one with the most launches -- config 5's table-building frames run another)
        k = sorted([n for n in s if is_frame_march(n)], key=lambda n: -max(v["n"] for v in s[n].values()))
        if not k:
            continue
        f, w = s[k[0]]["FETCH_SIZE"], s[k[0]]["WRITE_SIZE"]
        out = {"config": cfg, "tag": tag,
               "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 bench.py --config %s "
                          "--steps 3 --warmup 1 --no-cpu (separate passes); first launch of each kernel left out" % cfg,
               "kernel": k[0].replace("void ", ""), "launches": f["n"],
               "march_fetch_KB_per_launch": f["mean"], "march_write_KB_per_launch": w["mean"],
               "hbm_bytes_per_march_launch": (f["mean"] + w["mean"]) * 1024,
               "hbm_bytes_per_march_launch_upper": (2 * f["mean"] + w["mean"]) * 1024,
               "correction": "FETCH_SIZE x 1: 64 bytes tallied per random 1/8-byte gather (profiles/"
                             "r02_fetch_size_calibration.json); upper bound x 2, the guide's factor for wide coalesced "
                             "streaming reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; KB -> bytes x1024",
               "per_kernel": {n: {c: v for c, v in cs.items()} for n, cs in s.items()
                              if any(t in n for t in ("march", "rng", "raygen", "resolve"))}}
        # L2 counters of the same command (tools/pmc_tcc.sh, its own passes): hits / misses and the memory-side read requests
        tcc_dirs = [os.path.join(O, "pmc_%s_tcc" % cfg), os.path.join(O, "pmc_%s_tcc2" % cfg)]
        if all(os.path.isdir(d) for d in tcc_dirs):
            t = summarize(tcc_dirs, drop_first=True)
            # (config 5 has two march instances: the frames that build the tables record no traversed list -- the timed frames'
            # instance is the one with the most launches)
            tk = sorted([n for n in t if is_frame_march(n)], key=lambda n: -max(v["n"] for v in t[n].values()))
            if tk:
                c = {n: v["mean"] for n, v in t[tk[0]].items()}
                out["l2"] = {"command": "tools/pmc_tcc.sh: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum | TCC_EA0_RDREQ_sum "
                                        "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum (two passes), per march launch",
                             "counters": c,
                             "hit_rate": c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
                             "memory_side_read_bytes": c["TCC_EA0_RDREQ_sum"] * 64.0}
        json.dump(out, open(os.path.join(P, "pmc_%s.json" % cfg), "w"), indent=1)
        print(cfg, "HBM bytes per march launch: %.1f .. %.1f MB" % (out["hbm_bytes_per_march_launch"] / 1e6,
                                                                   out["hbm_bytes_per_march_launch_upper"] / 1e6))

if not pmc_only:
    # world flow, the one-ray-per-lane march of the same run, SQ summaries, loop statistics, shares
    for src, dst in (("bench_c3_world.json", "bench_c3_world.json"), ("prof_c3_world_bench.json", "prof_c3_world_bench.json"),
                     ("prof_c3_lanes_bench.json", "prof_c3_lanes_bench.json"), ("prof_c5_lanes_bench.json", "prof_c5_lanes_bench.json"),
                     ("pmc_c3_summary.txt", "sq_c3_summary.txt"), ("pmc_c5_summary.txt", "sq_c5_summary.txt"),
                     ("pmc_c3_lanes_summary.txt", "sq_c3_lanes_summary.txt"), ("diag_c3.txt", "diag_c3.txt"),
                     ("diag_c5.txt", "diag_c5.txt"), ("diag_c3_hist.txt", "diag_c3_hist.txt"), ("diag_c5_hist.txt", "diag_c5_hist.txt"),
                     ("share.txt", "share.txt"), ("policy_check.md", "policy_check.md"),
                     ("pmc_c3_tcc_summary.txt", "tcc_c3_summary.txt"), ("pmc_c5_tcc_summary.txt", "tcc_c5_summary.txt"),
                     ("diag_c3_ahead_hist.txt", "diag_c3_ahead_hist.txt"), ("diag_c5_lanes_w0_hist.txt", "diag_c5_lanes_hist.txt"),
                     ("diag_c5_lanes_w1_hist.txt", "diag_c5_lanes_ahead_hist.txt"),
                     ("prof_c3_ahead_bench.json", "prof_c3_ahead_bench.json"),
                     ("prof_c5_lanes_ahead_bench.json", "prof_c5_lanes_ahead_bench.json"),
                     ("bench_c3_reseed_table.json", "bench_c3_reseed_table.json"), ("prof_c3_reseed_bench.json", "prof_c3_reseed_bench.json"),
                     ("diag_c3_reseed.txt", "diag_c3_reseed.txt")):
        if os.path.exists(os.path.join(O, src)) and os.path.getsize(os.path.join(O, src)):
            shutil.copy(os.path.join(O, src), os.path.join(P, "%s_%s" % (tag, dst)))
    vrows = []
    # (a two-part profile run repeats the shipped march on part 2's box: prof2_*; else part 1's rows stand for it)
    both = os.path.isdir(os.path.join(O, "prof2_c3"))
    for sub, what in (("prof2_c3" if both else "prof_c3", "c3: ray pool (shipped)"), ("prof_c3_lanes", "c3: one ray per lane (VRT_POOL=0)"),
                      ("prof_c3_world", "c3 through Camera.set_world_scene + chunk_update (bench.py --world-flow)"),
                      ("prof_c3_reseed", "c3 without cached tables (bench.py --reseed): the lanes make their ray records"),
                      ("prof2_c5" if both else "prof_c5", "c5: ray pool (shipped)"), ("prof_c5_lanes", "c5: one ray per lane (VRT_POOL=0)"),
                      ("prof_c3_ahead", "c3: ray pool, look-ahead across chunk borders (VRT_WADDR=1, measured variant)"),
                      ("prof_c5_lanes_ahead", "c5: one ray per lane, look-ahead across chunk borders (VRT_WADDR=1 VRT_POOL=0)")):
        found = glob.glob(os.path.join(O, sub, "**", "*_kernel_stats.csv"), recursive=True)
        for f in ([max(found, key=os.path.getmtime)] if found else []):
            if sub not in ("prof_c3", "prof_c5", "prof2_c3", "prof2_c5"):
                shutil.copy(f, os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, sub.replace("prof_", ""))))
            for r in csv.DictReader(open(f)):
                if is_frame_march(r["Name"]):
                    calls = int(r["Calls"])
                    avg = (float(r["TotalDurationNs"]) - float(r["MaxNs"])) / max(1, calls - 1) / 1e6
                    vrows.append((what, r["Name"].replace("void ", "").split("(")[0], calls, avg))
    if vrows:
        with open(os.path.join(P, "%s_march_variants.md" % tag), "w") as fh:
            fh.write("# the frame's march, rocprofv3 --kernel-trace --stats, one MI355X, one run (tools/profile_all.sh)\n\n"
                     "Average launch duration without the first (cold) launch.\n\n"
                     "| run | kernel | launches | avg launch ms |\n|---|---|---|---|\n")
            for what, name, calls, avg in vrows:
                fh.write("| %s | `%s` | %d | %.3f |\n" % (what, name, calls, avg))
        print(open(os.path.join(P, "%s_march_variants.md" % tag)).read())

if not pmc_only:
    # which kernel source the set was measured with: the commit that last touched it (tests/test_bench_contract.py compares)
    import subprocess
    try:
        h = subprocess.run(["git", "log", "-1", "--format=%H", "--", "python_raytracer_amd/csrc", "include"], cwd=R,
                           capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--", "python_raytracer_amd/csrc", "include"], cwd=R,
                               capture_output=True, text=True, check=True).stdout.strip()
        open(os.path.join(P, "%s_commit.txt" % tag), "w").write(h + ("\n(uncommitted changes)\n" if dirty else "\n"))
    except Exception as e:  # (no git here)
        print("no commit record:", e)

# lookup variants (tools/lookup_variants.sh): the march rows of their kernel statistics
rows = []
for cfg in (() if pmc_only else ("c5", "c3")):
    for lk, what in ((0, "material bytes (shipped)"), (1, "occupancy words in registers"), (2, "8^3 occupancy bricks staged in LDS"),
                     ("roles", "wave roles: loader / finisher wave + 3 marching waves (material bytes)")):
        found = glob.glob(os.path.join(O, "lk_%s_%s" % (cfg, lk), "**", "*_kernel_stats.csv"), recursive=True)
        for f in ([max(found, key=os.path.getmtime)] if found else []):
            shutil.copy(f, os.path.join(P, "%s_lookup%s_%s_kernel_stats.csv" % (tag, lk, cfg)))
            for r in csv.DictReader(open(f)):
                if is_frame_march(r["Name"]):
                    calls = int(r["Calls"])
                    avg = (float(r["TotalDurationNs"]) - float(r["MaxNs"])) / max(1, calls - 1) / 1e6
                    rows.append((cfg, lk, what, r["Name"].replace("void ", "").split("(")[0], calls, avg))
if rows:
    with open(os.path.join(P, "%s_lookup_variants.md" % tag), "w") as fh:
        fh.write("# march_kernel measurement variants (VRT_LOOKUP, VRT_ROLES), 4-step speculation, rocprofv3 --kernel-trace --stats\n\n"
                 "`tools/lookup_variants.sh` on one MI355X; average launch duration without the first (cold) launch.\n"
                 "The kernel statistics files are `%s_lookup<variant>_<config>_kernel_stats.csv`.\n\n"
                 "| config | variant | kernel | launches | avg launch ms |\n|---|---|---|---|---|\n" % tag)
        for cfg, lk, what, name, calls, avg in rows:
            fh.write("| %s | %s: %s | `%s` | %d | %.3f |\n" % (cfg, lk, what, name, calls, avg))
    print(open(os.path.join(P, "%s_lookup_variants.md" % tag)).read())
