#!/usr/bin/env python3
"""Scheduling-knob sweep on the GPU box: runs bench.py once per environment setting and prints one line per run.
usage: sweep.py CONFIG "VAR=val VAR=val" ["VAR=val ..." ...]      (each argument is one run's environment)"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1]
steps = "3" if cfg == "c5" else "10"
for setting in sys.argv[2:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=")
        env[k] = v
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", cfg, "--no-cpu", "--steps", steps,
                          "--warmup", "2"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(setting, "FAILED", out.stderr[-300:], flush=True)
        continue
    d = json.loads(line[0])
    print("%-44s %s ms/frame %8.3f  march %8.3f  frac %.4f" % (setting, cfg, d["ms_per_step"], d["kernel_ms_per_step"]["march"],
                                                          d["roofline"]["frac"]), flush=True)
