#!/bin/bash
# One-off counter passes for kernel experiments (GPU box, through gpurun):
#   bash scripts/pmc_quick.sh NAME "ENV=.. ENV2=.." "COUNTERS PASS 1" "COUNTERS PASS 2" ...
# Each pass is its own rocprofv3 process with --kernel-trace only (never combined with other trace domains).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NAME=$1; ENVS=$2; shift 2
OUT=gpurun_out/pmcq_$NAME
rm -rf $OUT; mkdir -p $OUT
export $ENVS
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --distinct 16 ${BENCH_ARGS} > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 scripts/pmc_summary.py $OUT $OUT/traffic.json > $OUT/summary.csv 2>&1
head -${ROWS:-8} $OUT/summary.csv
rm -rf $OUT/p*/    # raw per-dispatch CSVs are large; the summary is what is kept
