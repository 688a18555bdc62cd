#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of the last gpurun from gpurun_out/ into profiles/ (tracked) and derive
profiles/pmc_<config>.json (HBM bytes per march launch, per MI355X_MICROARCH.md: FETCH_SIZE doubled on gfx950,
WRITE_SIZE as is, KB -> bytes).  usage: save_profiles.py TAG   e.g. r01_v5"""
import glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
os.makedirs(P, exist_ok=True)
for cfg in ("c3", "c5", "c2"):
    for f in glob.glob(os.path.join(O, "prof_%s" % cfg, "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, cfg)))
    for name in ("bench_%s.json" % cfg, "prof_%s_bench.json" % cfg, "bench_%s_rngcache.json" % cfg):
        src = os.path.join(O, name)
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(P, "%s_%s" % (tag, name)))
    dirs = [os.path.join(O, "pmc_fetch_%s" % cfg), os.path.join(O, "pmc_write_%s" % cfg)]
    if all(os.path.isdir(d) for d in dirs):
        s = summarize(dirs)
        k = [n for n in s if n.startswith("void march_kernel<false, false")]
        if not k:
            continue
        f, w = s[k[0]]["FETCH_SIZE"], s[k[0]]["WRITE_SIZE"]
        out = {"config": cfg, "tag": tag,
               "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 bench.py --config %s "
                          "--steps 1 --warmup 0 --no-cpu (separate passes)" % cfg,
               "kernel": k[0].replace("void ", ""), "launches": f["n"],
               "march_fetch_KB_per_launch": f["mean"], "march_write_KB_per_launch": w["mean"],
               "hbm_bytes_per_march_launch": (2 * f["mean"] + w["mean"]) * 1024,
               "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled (MI355X_MICROARCH.md, HBM); "
                             "WRITE_SIZE as is; KB -> bytes x1024. Uncalibrated for this kernel's 1/8-byte accesses.",
               "per_kernel": {n: {c: v for c, v in cs.items()} for n, cs in s.items()
                              if any(t in n for t in ("march", "rng", "raygen", "resolve"))}}
        json.dump(out, open(os.path.join(P, "pmc_%s.json" % cfg), "w"), indent=1)
        print(cfg, "HBM bytes per march launch: %.1f MB" % (out["hbm_bytes_per_march_launch"] / 1e6))
