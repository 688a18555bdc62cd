"""The committed bench lines (profiles/, written by bench.py on the MI355X box) carry every field of the bench
contract, and their derived fields are consistent."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the newest set that tools/save_profiles.py committed
TAG = max((os.path.basename(p).split("_bench_")[0] for p in glob.glob(os.path.join(ROOT, "profiles", "r*_v*_bench_c3.json"))),
          key=lambda t: [int(v) for v in t.replace("r", "").replace("v", "").split("_")])
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", TAG + "_bench_*.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_follows_the_contract(path):
    d = json.loads(open(path).read())
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["metric"].startswith("Mrays/s") and d["unit"] == "Mrays/s" and d["higher_is_better"] is True
    assert d["scaling"] in ("strong", "weak") and d["dtype"] == "f64" and "workload" in d["config"]
    rays = d["config"]["primary_rays"] + d["config"]["bounce_rays"]
    assert abs(d["value"] - rays / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-3 * r["achieved"]
    # counter traffic is what reached HBM / the fabric: a small world (config 2) is served from L2 and stays below the
    # algorithmic bytes, so only its presence and sign are part of the contract
    assert r["traffic"] is None or r["traffic"] > 0
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and c["sample"]


@pytest.mark.parametrize("cfg", ["c3", "c5", "c2"])
def test_bench_line_traffic_is_the_committed_counter_figure(cfg):
    """tools/profile_all.sh runs the counter passes before the bench lines, and tools/save_profiles.py derives
    profiles/pmc_<cfg>.json from the same counter files: the line's roofline.traffic is that file's figure."""
    d = json.loads(open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (TAG, cfg))).read())
    p = json.loads(open(os.path.join(ROOT, "profiles", "pmc_%s.json" % cfg)).read())
    assert p["tag"] == TAG and d["roofline"]["traffic"] == p["hbm_bytes_per_march_launch"]


def test_headline_line_has_a_cpu_baseline():
    d = json.loads(open(os.path.join(ROOT, "profiles", TAG + "_bench_c3.json")).read())
    assert "cpu_baseline" in d and d["n_gpus"] == 1 and "3840x2160" in d["config"]["workload"]


@pytest.mark.parametrize("cfg", ["c3", "c5", "c2"])
def test_rocprof_average_agrees_with_the_bench_line(cfg):
    """`rocprofv3 --kernel-trace --stats` of the same bench command: the dominant kernel's average duration agrees
    with the duration bench.py measured with HIP events (both committed by tools/save_profiles.py)."""
    import csv
    d = json.loads(open(os.path.join(ROOT, "profiles", "%s_prof_%s_bench.json" % (TAG, cfg))).read())
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (TAG, cfg)))))
    def frame_march(name):   # march_pool_kernel, or march_kernel<SPEC, RES, RECORD = false, LIST = false, ...>
        if name.startswith("void march_pool_kernel<"):
            return True
        if not name.startswith("void march_kernel<"):
            return False
        args = [a.strip() for a in name[len("void march_kernel<"):].split(">")[0].split(",")]
        return args[2] == "false" and args[3] == "false"
    march = [r for r in rows if frame_march(r["Name"])]
    # (config 5 runs two instances of the frame march: the frames that record `traversed` -- every timed one -- compare a
    # re-snap's key behind the voxel reads, the frames in which bench.py builds its tables record nothing)
    march.sort(key=lambda r: -float(r["TotalDurationNs"]))
    assert d["roofline"]["kernel"].split("<")[0] == march[0]["Name"].replace("void ", "").split("<")[0]
    assert len(march) <= 2 and int(march[0]["Calls"]) >= d["steps"]
    # rocprof also saw the untimed first frame (cold caches, tables being built: the one slowest call), which is left out;
    # the profiled command runs with --no-context, so every other launch is a timed frame
    calls = int(march[0]["Calls"])
    prof_ms = (float(march[0]["TotalDurationNs"]) - float(march[0]["MaxNs"])) / (calls - 1) / 1e6
    assert abs(prof_ms - d["roofline"]["avg_launch_ms"]) <= 0.05 * prof_ms
    assert float(march[0]["Percentage"]) > 50          # it is the dominant kernel


def test_profiles_are_of_the_kernel_source_in_the_tree():
    """The newest profile set was measured with the kernels that are in the tree: tools/save_profiles.py records the commit
    that last touched python_raytracer_amd/csrc and include/ when it copies a set into profiles/, and that is still the last
    commit that touched them (a kernel change needs a new tools/profile_all.sh run)."""
    import subprocess
    rec = os.path.join(ROOT, "profiles", TAG + "_commit.txt")
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here")
    assert os.path.exists(rec), "profiles/%s_commit.txt is missing: run tools/save_profiles.py %s" % (TAG, TAG)
    want = open(rec).read().split()
    p = subprocess.run(["git", "log", "-1", "--format=%H", "--", "python_raytracer_amd/csrc", "include"], cwd=ROOT,
                       capture_output=True, text=True)
    if p.returncode != 0 or not p.stdout.strip():
        pytest.skip("git not usable here")
    assert want and want[0] == p.stdout.strip() and len(want) == 1, (want, p.stdout.strip())
