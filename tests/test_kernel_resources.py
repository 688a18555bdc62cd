"""The frame march kernels must keep four waves per SIMD without vector spills: the compiler's own resource report
(-Rpass-analysis=kernel-resource-usage) of the shipped source, cross-compiled for gfx950 (no GPU needed).  A change that
pushes a frame kernel over 128 VGPRs costs tens of per cent on the GPU and would otherwise only show up in a bench run."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "python_raytracer_amd", "csrc", "vrt_kernels.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# mangled names of the kernels a frame runs at the BASELINE configurations
# (the last template arguments: W = the look-ahead crosses chunk borders, march_step_w -- the measured variant, VRT_WADDR=1;
# DEFER = the launch's traversed box has no settled bitmap and a re-snap's key is compared after the voxel reads went out;
# march_pool_kernel's TILE = rays handed out as square pixel tiles, an eighth of the window per XCD -- the measured variant, VRT_TILED=1)
FRAME_KERNELS = {
    "_Z17march_pool_kernelILi8ELi1ELi0ELb0ELb0ELb0EEv11MarchParams": "march_pool_kernel<8,1,0,false,false,false> (config 3)",
    "_Z17march_pool_kernelILi8ELi0ELi1ELb0ELb1ELb0EEv11MarchParams": "march_pool_kernel<8,0,1,false,true,false> (config 5: keys compared late)",
    "_Z12march_kernelILi8ELi1ELb0ELb0ELi0ELi0ELb0ELb0ELi0EEv11MarchParams": "march_kernel<8,1,false,false,0,0,false,false> (config 2)",
    "_Z12march_kernelILi8ELi0ELb0ELb0ELi0ELi2ELb0ELb1ELi0EEv11MarchParams": "march_kernel<8,0,false,false,0,2,false,true>",
    "_Z17march_pool_kernelILi8ELi0ELi1ELb0ELb0ELb0EEv11MarchParams": "march_pool_kernel<8,0,1,false,false,false>",
    "_Z17march_pool_kernelILi8ELi1ELi0ELb0ELb1ELb0EEv11MarchParams": "march_pool_kernel<8,1,0,false,true,false>",
    "_Z12march_kernelILi8ELi0ELb0ELb0ELi0ELi0ELb0ELb0ELi0EEv11MarchParams": "march_kernel<8,0,false,false,0,0,false,false>",
    "_Z17march_pool_kernelILi8ELi1ELi0ELb1ELb0ELb0EEv11MarchParams": "march_pool_kernel<8,1,0,true,false,false> (look-ahead variant)",
    "_Z17march_pool_kernelILi8ELi0ELi1ELb1ELb0ELb0EEv11MarchParams": "march_pool_kernel<8,0,1,true,false,false> (look-ahead variant)",
    "_Z12march_kernelILi8ELi1ELb0ELb0ELi0ELi0ELb1ELb0ELi0EEv11MarchParams": "march_kernel<8,1,false,false,0,0,true,false> (look-ahead variant)",
    "_Z17march_pool_kernelILi8ELi0ELi1ELb0ELb1ELb1EEv11MarchParams": "march_pool_kernel<8,0,1,false,true,true> (tiled hand-out variant)",
    "_Z17march_pool_kernelILi8ELi1ELi0ELb0ELb1ELb1EEv11MarchParams": "march_pool_kernel<8,1,0,false,true,true> (tiled hand-out variant)",
}
# frames without a cached ray table: the lanes make their own ray records (take_ray, PERPIX 3) -- the sines and cosines of the
# lens quaternion bring constants (scalar registers spilled around them, in the refill only) and the out-of-line slow path's
# call frame, but must not cost the march its four waves
RAYGEN_KERNELS = {
    "_Z17march_pool_kernelILi8ELi1ELi3ELb0ELb0ELb0EEv11MarchParams": "march_pool_kernel<8,1,3,false,false,false> (config 3 --reseed)",
    "_Z17march_pool_kernelILi8ELi0ELi3ELb0ELb0ELb0EEv11MarchParams": "march_pool_kernel<8,0,3,false,false,false>",
    "_Z17march_pool_kernelILi8ELi1ELi3ELb0ELb1ELb0EEv11MarchParams": "march_pool_kernel<8,1,3,false,true,false>",
    "_Z17march_pool_kernelILi8ELi0ELi3ELb0ELb1ELb0EEv11MarchParams": "march_pool_kernel<8,0,3,false,true,false>",
    "_Z12march_kernelILi8ELi1ELb0ELb0ELi0ELi3ELb0ELb0ELi0EEv11MarchParams": "march_kernel<8,1,false,false,0,3,false,false> (config 2 --reseed)",
    "_Z12march_kernelILi8ELi0ELb0ELb0ELi0ELi3ELb0ELb0ELi0EEv11MarchParams": "march_kernel<8,0,false,false,0,3,false,false>",
    "_Z12march_kernelILi8ELi1ELb0ELb0ELi0ELi3ELb0ELb1ELi0EEv11MarchParams": "march_kernel<8,1,false,false,0,3,false,true>",
    "_Z12march_kernelILi8ELi0ELb0ELb0ELi0ELi3ELb0ELb1ELi0EEv11MarchParams": "march_kernel<8,0,false,false,0,3,false,true>",
}


@pytest.fixture(scope="module")
def report(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("res") / "vrt.o"
    p = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "--cuda-device-only", "-c",
                        "-Rpass-analysis=kernel-resource-usage", SRC, "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = {}
    for b in re.split(r"(?=remark: Function Name: )", p.stderr):
        m = re.match(r"remark: Function Name: (\S+)", b)
        if m:
            blocks[m.group(1)] = {k: int(v) for k, v in re.findall(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", b)}
    return blocks


@pytest.mark.parametrize("name", sorted(FRAME_KERNELS))
def test_frame_kernels_keep_four_waves_without_vector_spills(report, name):
    assert name in report, "kernel %s not in the library any more" % FRAME_KERNELS[name]
    r = report[name]
    assert r["VGPRs Spill"] == 0, (FRAME_KERNELS[name], r)
    assert r["Occupancy [waves/SIMD]"] >= 4, (FRAME_KERNELS[name], r)
    assert r["VGPRs"] <= 128 and r["AGPRs"] == 0, (FRAME_KERNELS[name], r)
    # 16 bytes: the call frame of the pow slow path; anything more is ray state in scratch memory
    assert r["ScratchSize [bytes/lane]"] <= 16, (FRAME_KERNELS[name], r)
    # scalar registers spilled to vector lanes cost a v_readlane per use -- VALU work.  Round 3's kernels had 21 (ray pool) /
    # 6-9 (one ray per lane); the re-snap's wave-uniform switches now travel as one re-read word (MarchParams::snap_flags)
    # instead of as hoisted 64-bit lane masks, and the ray pool's march step re-reads its arguments (VRT_FRESH_MARCH_POOL):
    # 6 / 4 in the shipped kernels (the measured variants -- look-ahead, tiled hand-out -- up to 24)
    assert r["SGPRs Spill"] <= (24 if "variant" in FRAME_KERNELS[name] else 8), (FRAME_KERNELS[name], r)


@pytest.mark.parametrize("name", sorted(RAYGEN_KERNELS))
def test_kernels_that_make_their_ray_records_keep_four_waves(report, name):
    assert name in report, "kernel %s not in the library any more" % RAYGEN_KERNELS[name]
    r = report[name]
    assert r["VGPRs Spill"] == 0 and r["Occupancy [waves/SIMD]"] >= 4 and r["VGPRs"] <= 128 and r["AGPRs"] == 0, (RAYGEN_KERNELS[name], r)
    assert r["ScratchSize [bytes/lane]"] <= 64 and r["SGPRs Spill"] <= 112, (RAYGEN_KERNELS[name], r)
