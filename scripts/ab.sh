#!/bin/bash
# A/B of one environment knob on one box: scripts/ab.sh KNOB "v1 v2 ..." [bench args]   -> gpurun_out/ab_<KNOB>_<v>.json, prints the
# headline and the per-kernel milliseconds (bench.py --no-cpu-baseline --no-variants; three runs per value, interleaved).
knob=$1; vals=$2; shift 2
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in $vals; do
    env $knob=$v python bench.py --no-cpu-baseline --no-variants --distinct 32 "$@" > gpurun_out/ab_${knob}_${v}_$rep.json 2> gpurun_out/ab_${knob}_${v}_$rep.err || { echo "run failed: $knob=$v"; tail -5 gpurun_out/ab_${knob}_${v}_$rep.err; exit 1; }
    python - "$knob=$v" gpurun_out/ab_${knob}_${v}_$rep.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = r["roofline"]["kernels_ms_per_step"]
print(sys.argv[1], "value %.0f ms/step %.3f |" % (r["value"], r["ms_per_step"]), " ".join("%s %.3f" % (n, v) for n, v in list(k.items())[:6]), flush=True)
PY
  done
done
