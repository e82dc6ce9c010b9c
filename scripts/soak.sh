#!/bin/bash
# Longer validation of ONE build than the test suite runs (r05): run-to-run determinism at three shapes, the metric bar on 64 pairs each
# (repeatability / MHA and pose AUC), a 2000-step bench line.  scripts/soak.sh <round>  ->  gpurun_out/<round>/soak_*.txt|json
R=${1:-r05}; O=gpurun_out/$R; mkdir -p $O
sha256sum keypoint_bench_amd/libkpb.so | cut -c1-12 > $O/soak_build.txt
{ python scripts/determinism_probe.py 1216 1600 300 dense 4 | tail -1
  python scripts/determinism_probe.py 800 1216 200 dense 8 | tail -1
  python scripts/determinism_probe.py 480 640 60 dense 128 | tail -1
  python scripts/determinism_probe.py 480 640 200 sparse 128 | tail -1; } > $O/soak_determinism.txt 2>&1
cat $O/soak_determinism.txt
python scripts/metric_sweep.py 64 $O/soak_metric_sweep_64.json > $O/soak_metric_sweep_64.log 2>&1; tail -2 $O/soak_metric_sweep_64.log
python scripts/metric_sweep.py 64 --auc $O/soak_metric_sweep_auc_64.json > $O/soak_metric_sweep_auc_64.log 2>&1; tail -2 $O/soak_metric_sweep_auc_64.log
python bench.py --no-cpu-baseline --no-variants --steps 2000 --warmup 20 2>/dev/null | tail -1 > $O/soak_bench_2000_steps.json
python - $O/soak_bench_2000_steps.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read())
print("2000 steps: value %.0f sustained %.0f ms/step %.3f build %s" % (r["value"], r["value_sustained"], r["ms_per_step"], r["config"]["build"]["lib_sha256"]))
PY
