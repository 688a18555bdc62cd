"""The march's scheduling defaults away from the launch sizes they were tuned on (BASELINE configs 2, 3, 5): for the
in-between configurations x1..x3 of bench.py, the frame kernel the defaults pick against the other one and against
neighbouring knob settings.  Every run must give the same image.  usage (GPU box): python tools/policy_check.py > table"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [("defaults", {}),
        ("one ray per lane", {"VRT_POOL": "0"}),
        ("pool forced", {"VRT_POOL_MIN_RAYS": "0"}),
        ("pool, t_hit 48 t_end 48", {"VRT_POOL_MIN_RAYS": "0", "VRT_POOL_T_HIT": "48", "VRT_POOL_T_END": "48"}),
        ("pool, t_hit 32 t_end 60", {"VRT_POOL_MIN_RAYS": "0", "VRT_POOL_T_HIT": "32", "VRT_POOL_T_END": "60"}),
        ("pool, iters 5", {"VRT_POOL_MIN_RAYS": "0", "VRT_POOL_ITERS": "5"}),
        ("chunk 128", {"VRT_CHUNK": "128"}),
        ("chunk 512", {"VRT_CHUNK": "512"})]
print("| config | rays | setting | kernel | march ms | frame ms | image |")
print("|---|---|---|---|---|---|---|")
for cfg in sys.argv[1:] or ["x1", "x2", "x3"]:
    for name, extra in RUNS:
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "10", "--warmup", "3",
                              "--no-cpu", "--no-context"], env=env, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print("|", cfg, "| |", name, "| FAILED |", out.stderr[-200:].replace("\n", " "), "| | |")
            continue
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("| %s | %d | %s | %s | %.3f | %.3f | %s |" % (cfg, d["config"]["primary_rays"], name, d["roofline"]["kernel"],
                                                         d["kernel_ms_per_step"]["march"], d["ms_per_step"],
                                                         d["config"]["image_sha256"][:8]), flush=True)
