#!/bin/bash
# One or more quick PMC passes over a short bench run, per-kernel averages printed:  bash scripts/pmc_quick.sh <tag> "<counters>" ["<counters>" ...]
# (extra bench.py arguments through BENCH_ARGS; each pass is its own rocprofv3 process, --kernel-trace only)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants $BENCH_ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python scripts/pmc_summary.py $OUT > $OUT/summary.csv 2>&1
python - <<PY
import csv
rows = list(csv.reader(open("$OUT/summary.csv")))
hdr = rows[0]
for r in rows[1:12]:
    print("  ".join("%s=%s" % (h, v) for h, v in zip(hdr, r) if v))
PY
