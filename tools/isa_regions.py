#!/usr/bin/env python3
"""Static instruction counts of a march kernel between the -DVRT_ISA_MARK comments (vrt_kernels.hip: VRT_MARK).

    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DVRT_ISA_MARK --cuda-device-only -S \
        python_raytracer_amd/csrc/vrt_kernels.hip -o /tmp/mark.s
    python tools/isa_regions.py /tmp/mark.s [_Z17march_pool_kernelILi8ELi1ELi0EEv11MarchParams]

Regions are taken in layout order (the compiler may move a block; the sums are what matters), one row per marker:
VALU / SALU / LDS / VMEM / branch instructions, and the VALU cycle weight (64-bit and transcendental operations issue
over more cycles: profiles/r02_microbench_valu.json).
"""
import re
import sys
from collections import OrderedDict


def klass(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def weight(op):
    """issue cycles of a VALU instruction relative to a 32-bit one (4 waves per SIMD, measured)"""
    if op in ("v_rcp_f64_e32", "v_rsq_f64_e32", "v_sqrt_f64_e32", "v_rcp_f64", "v_div_scale_f64", "v_div_fmas_f64",
              "v_div_fixup_f64"):
        return 4.0 if op.startswith(("v_rcp", "v_rsq", "v_sqrt")) else 2.0
    if "f64" in op or "b64" in op or "u64" in op or "i64" in op:
        return 1.35
    return 1.0


def main():
    path = sys.argv[1]
    fn = sys.argv[2] if len(sys.argv) > 2 else "_Z17march_pool_kernelILi8ELi1ELi0EEv11MarchParams"
    lines = open(path).read().split("\n")
    start = lines.index(fn + ":" + " ; @" + fn) if (fn + ": ; @" + fn) in lines else None
    if start is None:
        for i, l in enumerate(lines):
            if l.startswith(fn + ":"):
                start = i
                break
    regions = OrderedDict()
    cur = "prologue"
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r";\s*@@(\w+)", t)
        if m:
            cur = m.group(1)
            continue
        if not t or t.startswith((";", ".", "#")) or t.endswith(":"):
            continue
        op = t.split()[0]
        k = klass(op)
        r = regions.setdefault(cur, dict(valu=0, salu=0, lds=0, vmem=0, branch=0, wait=0, smem=0, other=0, vw=0.0))
        r[k] += 1
        if k == "valu":
            r["vw"] += weight(op)
    print("%-12s %6s %6s %5s %5s %6s %5s %5s %8s" % ("region", "valu", "salu", "lds", "vmem", "branch", "wait", "smem", "valu_w"))
    tot = dict()
    for name, r in regions.items():
        print("%-12s %6d %6d %5d %5d %6d %5d %5d %8.0f" % (name, r["valu"], r["salu"], r["lds"], r["vmem"], r["branch"],
                                                           r["wait"], r["smem"], r["vw"]))
        for k, v in r.items():
            tot[k] = tot.get(k, 0) + v
    print("%-12s %6d %6d %5d %5d %6d %5d %5d %8.0f" % ("total", tot["valu"], tot["salu"], tot["lds"], tot["vmem"],
                                                       tot["branch"], tot["wait"], tot["smem"], tot["vw"]))


if __name__ == "__main__":
    main()
