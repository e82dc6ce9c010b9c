#!/bin/bash
# one bench run per library given (measurement-only builds of scripts/hack_build.py): scripts/knock.sh "bench args" lib1.so lib2.so ...
# The libraries are loaded through KPB_LIB_PATH (keypoint_bench_amd/_lib.py): the tree's own libkpb.so is never overwritten.
args=$1; shift
for lib in "$@"; do
  KPB_LIB_PATH=$(realpath $lib) python bench.py --no-cpu-baseline --no-variants --distinct 32 $args > gpurun_out/knock.json 2> gpurun_out/knock.err || { echo "run failed: $lib"; tail -3 gpurun_out/knock.err; }
  python - "$lib" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/knock.json").read().strip().splitlines()[-1])
k = r["roofline"]["kernels_ms_per_step"]
print(sys.argv[1].split("libkpb_")[-1], "ms/step %.3f |" % r["ms_per_step"], " ".join("%s %.3f" % (n, v) for n, v in list(k.items())[:7]), flush=True)
PY
done
