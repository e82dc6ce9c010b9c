"""Builds keypoint_bench_amd/libkpb.so (HIP, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU
box with the repository snapshot.  -ffp-contract=off: the parity contract needs un-fused float
arithmetic where the reference's CPU code has none (kernels that want FMAs call fmaf explicitly).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libkpb.so")
SOURCES = ["api.hip", "detect.hip", "match.hip", "net_api.hip", "alike.hip", "convnet.hip", "lightglue.hip", "covis.hip", "lk.hip", "preprocess.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-result", "-fvisibility=hidden"]


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "kpb.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = ["hipcc"] + FLAGS + ["-o", SO] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)
