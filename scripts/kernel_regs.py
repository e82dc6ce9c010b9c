#!/usr/bin/env python3
"""Register / LDS / spill figures of the kernels in libkpb.so whose (mangled) name contains one of the given substrings:
    python scripts/kernel_regs.py nms_sweep_r alike_block   (llvm-objdump --offloading + llvm-readelf --notes; no GPU needed)"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = "/opt/rocm/lib/llvm/bin"


def main():
    so = os.environ.get("KPB_LIB_PATH") or os.path.join(ROOT, "keypoint_bench_amd", "libkpb.so")
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, "lib.so")
        shutil.copy(so, local)
        subprocess.run([os.path.join(TOOLS, "llvm-objdump"), "--offloading", local], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in sorted(glob.glob(local + ".*gfx950*")):
            txt = subprocess.run([os.path.join(TOOLS, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name or (pats and not any(p in name.group(1) for p in pats)):
                    continue
                f = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
                print("%-90s vgpr %s sgpr %s lds %s scratch %s vgpr_spill %s" % (name.group(1)[:90], f("vgpr_count"), f("sgpr_count"), f("group_segment_fixed_size"),
                                                                                  f("private_segment_fixed_size"), f("vgpr_spill_count")))


if __name__ == "__main__":
    main()
