#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: usage  pmc_summary.py DIR [DIR ...]
Prints, per kernel name and counter, the number of dispatches and the mean / sum of the counter."""
import collections
import csv
import glob
import json
import sys


def summarize(dirs):
    agg = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[(r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out = collections.defaultdict(dict)
    for (k, c), v in sorted(agg.items()):
        out[k][c] = {"n": len(v), "mean": sum(v) / len(v), "sum": sum(v)}
    return out


if __name__ == "__main__":
    out = summarize(sys.argv[1:])
    for k, cs in out.items():
        if not any(t in k for t in ("march", "rng", "resolve", "raygen")):
            continue
        print(k)
        for c, s in cs.items():
            print("   %-28s n=%-4d mean=%-16.1f sum=%.4g" % (c, s["n"], s["mean"], s["sum"]))
    json.dump(out, open("/dev/stdout" if len(sys.argv) < 2 else "/dev/null", "w"))
