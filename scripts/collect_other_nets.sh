#!/bin/bash
# Bench lines of the other §8(a) rows (N2..N5), run on the GPU box:  bash scripts/collect_other_nets.sh r01
R=${1:-r01}
mkdir -p gpurun_out/$R
for spec in "superpoint brute_force" "xfeat brute_force" "disk brute_force" "superpoint lightglue" "disk lightglue"; do
  set -- $spec
  timeout -k 10 300 python bench.py --model $1 --matcher $2 --no-cpu-baseline > gpurun_out/$R/bench_$1_$2.json 2> gpurun_out/$R/bench_$1_$2.err || echo "$spec failed"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$R/bench_*_*.json")):
    try:
        d = json.load(open(f))
        r = d["roofline"]
        print(f.split("/")[-1], d["value"], d["unit"], r["kernel"], r["frac"], {k: v for k, v in list(r["kernels_ms_per_step"].items())[:6]})
    except Exception as e:
        print(f, "unreadable", e)
PY
