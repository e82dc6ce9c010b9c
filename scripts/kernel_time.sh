#!/bin/bash
# Mean duration of the kernels whose name contains SUBSTR, for each library given, from a rocprofv3 kernel trace of a few steps on ONE box (measurement
# builds of scripts/hack_build.py beside the tree's library; loaded through KPB_LIB_PATH):  scripts/kernel_time.sh SUBSTR "bench args" lib1.so lib2.so ...
export TMPDIR=/tmp
sub=$1; bargs=$2; shift 2
for lib in "$@"; do
  d=gpurun_out/kt_$(basename $lib .so)
  rm -rf $d
  KPB_LIB_PATH=$(realpath $lib) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py $bargs --steps 6 --warmup 2 --no-cpu-baseline --no-variants --distinct 32 > $d.log 2>&1
  python3 - "$lib" $d "$sub" <<'PY'
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[2] + "/*/*kernel_trace.csv")
if not f:
    print(sys.argv[1], "no trace"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if sys.argv[3] in r["Kernel_Name"]]
print("%-28s %s: %.3f ms (n=%d, min %.3f, median %.3f)" % (sys.argv[1].split("libkpb_")[-1], sys.argv[3], st.mean(d) if d else 0, len(d), min(d) if d else 0, st.median(d) if d else 0), flush=True)
PY
done
