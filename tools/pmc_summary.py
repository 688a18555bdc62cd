#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: usage  pmc_summary.py DIR [DIR ...]
Prints, per kernel name and counter, the number of dispatches and the mean / sum of the counter.  With
drop_first, the first dispatch of every kernel (cold caches, tables being built) is left out of the mean."""
import collections
import csv
import glob
import os
import sys


def summarize(dirs, drop_first=False):
    agg = collections.defaultdict(list)
    for d in dirs:
        # (gpurun merges a run's files into gpurun_out/: an older run's file may still lie beside the new one)
        found = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
        for f in ([max(found, key=os.path.getmtime)] if found else []):
            rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
            for r in rows:
                agg[(r["Kernel_Name"].split("(")[0][:56], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out = collections.defaultdict(dict)
    for (k, c), v in sorted(agg.items()):
        if drop_first and len(v) > 1:
            v = v[1:]
        out[k][c] = {"n": len(v), "mean": sum(v) / len(v), "sum": sum(v)}
    return out


if __name__ == "__main__":
    out = summarize(sys.argv[1:], drop_first=True)
    print("# per-dispatch means without each kernel's first dispatch (cold caches, tables being built)")
    for k, cs in out.items():
        if not any(t in k for t in ("march", "rng", "resolve", "raygen")):
            continue
        print(k)
        for c, s in cs.items():
            print("   %-28s n=%-4d mean=%-16.1f sum=%.4g" % (c, s["n"], s["mean"], s["sum"]))
