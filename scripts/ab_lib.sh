#!/bin/bash
# A/B of two BUILDS of libkpb.so on one box: scripts/ab_lib.sh base.so new.so [bench args] (three interleaved runs each; loaded through KPB_LIB_PATH, the tree's library is never overwritten)
base=$1; new=$2; shift 2
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then lib=$base; else lib=$new; fi
    KPB_LIB_PATH=$(realpath $lib) python bench.py --no-cpu-baseline --no-variants --distinct 32 "$@" > gpurun_out/abl_${v}_$rep.json 2> gpurun_out/abl_${v}_$rep.err || { echo "run failed: $v"; tail -5 gpurun_out/abl_${v}_$rep.err; exit 1; }
    python - "$v" gpurun_out/abl_${v}_$rep.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = r["roofline"]["kernels_ms_per_step"]
print(sys.argv[1], "value %.0f ms/step %.3f |" % (r["value"], r["ms_per_step"]), " ".join("%s %.3f" % (n, v) for n, v in list(k.items())[:5]), flush=True)
PY
  done
done
