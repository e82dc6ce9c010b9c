#!/usr/bin/env python3
"""MEASUREMENT-ONLY builds of libkpb.so that never touch the tree: scripts/hack_build.py NAME PATCH.py

Copies keypoint_bench_amd/ (sources, build.py, isa_fixup.py and the objects already built) and include/ to /tmp/kpb_hack/NAME, calls
patch(csrc_dir) of PATCH.py on the COPY, builds there and leaves the library as scripts/_bin/libkpb_NAME.so (git-ignored, travels to the
GPU box) for scripts/ab_lib.sh.  PATCH.py helpers: sub(path, old, new, count=1) fails when `old` is not found exactly `count` times.
NAME `base` with no patch = the tree as it is.  (r05: two hack builds edited in place cost an afternoon's uncommitted work.)
"""
import importlib.util
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sub(path, old, new, count=1):
    text = open(path).read()
    n = text.count(old)
    if n != count:
        raise SystemExit("%s: expected %d occurrence(s) of %r, found %d" % (path, count, old[:60], n))
    open(path, "w").write(text.replace(old, new))


def main():
    name = sys.argv[1]
    patch = sys.argv[2] if len(sys.argv) > 2 else None
    work = os.path.join("/tmp/kpb_hack", name)
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    shutil.copytree(os.path.join(ROOT, "keypoint_bench_amd"), os.path.join(work, "keypoint_bench_amd"), symlinks=True,
                    ignore=shutil.ignore_patterns("__pycache__", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "include"))
    if patch:
        spec = importlib.util.spec_from_file_location("hack_patch", patch)
        mod = importlib.util.module_from_spec(spec)
        mod.sub = sub
        spec.loader.exec_module(mod)
        mod.patch(os.path.join(work, "keypoint_bench_amd", "csrc"))
    subprocess.run([sys.executable, os.path.join(work, "keypoint_bench_amd", "build.py")], check=True, cwd=work)
    out = os.path.join(ROOT, "scripts", "_bin", "libkpb_%s.so" % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    shutil.copy(os.path.join(work, "keypoint_bench_amd", "libkpb.so"), out)
    print(out)


if __name__ == "__main__":
    main()
