#!/bin/bash
# Bench lines and per-kernel PMC summaries of the other 8(a) rows (N2..N5), run on the GPU box:  bash scripts/collect_other_nets.sh r03 [pmc]
# With a second argument the PMC passes (scripts/prof_pmc.sh: SQ, matrix-pipe, FETCH_SIZE, WRITE_SIZE, GRBM) run for SuperPoint,
# DISK + LightGlue and XFeat as well.
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$R
for spec in "superpoint brute_force" "xfeat brute_force" "disk brute_force" "superpoint lightglue" "disk lightglue"; do
  set -- $spec
  timeout -k 10 300 python bench.py --model $1 --matcher $2 --no-cpu-baseline > gpurun_out/$R/bench_$1_$2.json 2> gpurun_out/$R/bench_$1_$2.err || echo "$spec failed"
done
# the opt-in f16 attention of LightGlue (what the reference runs on a GPU, lightglue.py:129-134), reported separately
for net in superpoint disk; do
  timeout -k 10 300 python bench.py --model $net --matcher lightglue --lg-attention f16 --no-cpu-baseline --no-variants > gpurun_out/$R/bench_${net}_lightglue_f16attn.json 2> gpurun_out/$R/bench_${net}_lightglue_f16attn.err || echo "$net lightglue f16 failed"
done
if [ -n "$2" ] || [ -n "$PMC" ]; then
  bash scripts/prof_pmc.sh ${R}_superpoint --model superpoint > gpurun_out/$R/pmc_superpoint.txt 2>&1
  bash scripts/prof_pmc.sh ${R}_disk_lightglue --model disk --matcher lightglue > gpurun_out/$R/pmc_disk_lightglue.txt 2>&1
  bash scripts/prof_pmc.sh ${R}_xfeat --model xfeat > gpurun_out/$R/pmc_xfeat.txt 2>&1
fi
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
