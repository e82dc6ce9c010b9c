#!/usr/bin/env python3
"""Which kernels are near NO roof?  scripts/kernel_scan.py [round]  reads profiles/<round>_pmc_per_kernel_*.csv (scripts/prof_pmc.sh) and prints, per
kernel with more than 0.15 ms per profiled run: average time, counter traffic (2 FETCH_SIZE + WRITE_SIZE), the TB/s that is, the share of the SIMDs'
vector issue slots its instructions take (SQ_INSTS_VALU x 4 cycles over GRBM_GUI_ACTIVE x 1024 SIMDs) and the share of its wave-cycles spent waiting.
A kernel far below 5 TB/s AND far below 1.0 of vector issue is waiting for latency or for itself: r05 found alike_score_lin (issue-bound, 505
instructions per pixel), alike_desc_at (16 KB from L2 per keypoint) and DISK's serial make_xf this way."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd + "_pmc_per_kernel_*.csv"))):
        print("==", os.path.basename(path))
        for r in csv.DictReader(open(path)):
            f = lambda k: float(r.get(k, 0) or 0)
            us, calls = f("avg_us"), f("calls")
            if us * calls < 150:
                continue
            gb = (2 * f("FETCH_SIZE") + f("WRITE_SIZE")) * 1024 / 1e9
            gui = f("GRBM_GUI_ACTIVE") / 8.0                     # summed over the 8 XCDs
            issue = f("SQ_INSTS_VALU") * 4 / (gui * 1024) if gui else 0.0
            print("%-44s %9.1f us x%4d %7.2f GB %5.2f TB/s  vector issue %4.2f  waiting %4.2f  valu/mfma %s"
                  % (r["kernel"][:44], us, calls, gb, gb / (us * 1e-6) / 1e3 if us else 0.0, issue, f("SQ_WAIT_ANY") / max(f("SQ_WAVE_CYCLES"), 1.0),
                     "%.1f" % (f("SQ_INSTS_VALU") / f("SQ_INSTS_MFMA")) if f("SQ_INSTS_MFMA") else "-"))


if __name__ == "__main__":
    main()
