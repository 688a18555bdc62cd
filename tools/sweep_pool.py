"""Sweep of march_pool_kernel's scheduling knobs (each setting in its own process: the knobs are read once).
usage (GPU box): python tools/sweep_pool.py c3|c5 'T_HIT,T_END,SWAP_MIN,REFILL_MIN,KEEP,ITERS' ...   ('-' = default)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1]
names = ["VRT_POOL_T_HIT", "VRT_POOL_T_END", "VRT_POOL_SWAP_MIN", "VRT_POOL_REFILL_MIN", "VRT_POOL_KEEP", "VRT_POOL_ITERS"]
steps, warm = ("3", "1") if cfg == "c5" else ("10", "2")
for combo in sys.argv[2:]:
    env = dict(os.environ)
    for n, v in zip(names, combo.split(",")):
        if v != "-":
            env[n] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", steps, "--warmup", warm,
                          "--no-cpu", "--no-context"], env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        print(combo, "FAILED", out.stderr[-300:])
        continue
    d = json.loads(out.stdout.strip().splitlines()[-1])
    print("%s %-22s frame %.4f ms  march %.4f ms  %s" % (cfg, combo, d["ms_per_step"], d["kernel_ms_per_step"]["march"], d["config"]["image_sha256"][:8]), flush=True)
