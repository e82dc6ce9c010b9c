"""Builds keypoint_bench_amd/libkpb.so (HIP, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU
box with the repository snapshot.  -ffp-contract=off: the parity contract needs un-fused float
arithmetic where the reference's CPU code has none (kernels that want FMAs call fmaf explicitly).

Each translation unit is compiled to its own object (only the stale ones, in parallel) and the objects
are linked into the shared library.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
SO = os.path.join(HERE, "libkpb.so")
SOURCES = ["api.hip", "detect.hip", "match.hip", "net_api.hip", "alike.hip", "convnet.hip", "lightglue.hip", "covis.hip", "lk.hip",
           "preprocess.hip", "geometry.hip"]
# -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary vector registers.  gfx950's register file is unified, so accumulation
# registers buy no occupancy, and every value that crosses between them and the vector ALUs costs a v_accvgpr_read / _write: 240 of
# them in lg_flash_h (192 -> 145 registers without), 24 in alike_block2, 8 in alike_block1_h (r03, found in the ISA).
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-result", "-fvisibility=hidden", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "kpb.h"),
                                                                                     os.path.abspath(__file__)]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    return _stale(SO, srcs + _headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append(["hipcc"] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES])
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)
