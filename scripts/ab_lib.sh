#!/bin/bash
# A/B of two BUILDS of libkpb.so on one box: scripts/ab_lib.sh base.so new.so [bench args] (three interleaved runs each)
base=$1; new=$2; shift 2
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then cp $base keypoint_bench_amd/libkpb.so; else cp $new keypoint_bench_amd/libkpb.so; fi
    python bench.py --no-cpu-baseline --no-variants --distinct 32 "$@" > gpurun_out/abl_${v}_$rep.json 2> gpurun_out/abl_${v}_$rep.err || { echo "run failed: $v"; tail -5 gpurun_out/abl_${v}_$rep.err; exit 1; }
    python - "$v" gpurun_out/abl_${v}_$rep.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = r["roofline"]["kernels_ms_per_step"]
print(sys.argv[1], "value %.0f ms/step %.3f |" % (r["value"], r["ms_per_step"]), " ".join("%s %.3f" % (n, v) for n, v in list(k.items())[:5]), flush=True)
PY
  done
done
cp $new keypoint_bench_amd/libkpb.so
