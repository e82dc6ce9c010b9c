#!/bin/bash
# Duration of sweep 0 of nms_sweep_r<6> (the launches longer than 0.5 ms) for each library given, from a kernel trace of a few sparse steps -- the knock-out builds of
# scripts/hack_build.py give wrong maps (the status check may re-run sweeps): only the first sweep's time (and the tail's) is read.  scripts/nms_knock.sh lib1.so lib2.so ...
export TMPDIR=/tmp
for lib in "$@"; do
  d=gpurun_out/nmsko_$(basename $lib .so)
  rm -rf $d
  KPB_LIB_PATH=$(realpath $lib) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --sparse --steps 6 --warmup 2 --no-cpu-baseline --no-variants --distinct 32 > $d.log 2>&1
  python3 - "$lib" $d <<'PY'
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[2] + "/*/*kernel_trace.csv")
if not f:
    print(sys.argv[1], "no trace"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "nms_sweep_r" in r["Kernel_Name"]]
big = [x for x in d if x > 0.5]
tl = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "nms_tail" in r["Kernel_Name"]]
b1 = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "alike_block1_h" in r["Kernel_Name"]]
print("%-28s sweep 0: %.3f ms (n=%d, min %.3f)   all sweeps %d   tail %.3f   block1 %.3f" % (sys.argv[1].split("libkpb_")[-1], st.mean(big) if big else 0, len(big), min(big) if big else 0, len(d), st.mean(tl) if tl else 0, st.mean(b1)), flush=True)
PY
done
