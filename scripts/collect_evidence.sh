#!/bin/bash
# Round evidence, run on the GPU box through gpurun:  bash scripts/collect_evidence.sh r01
# Order matters: the PMC passes come first and their per-kernel HBM traffic is put where bench.py looks for it
# (profiles/pmc_traffic_*.json), so the bench line recorded under rocprofv3 --stats carries the traffic of THIS build.
# Leaves everything under gpurun_out/$1/; scripts/publish_evidence.py copies the summaries into profiles/ (tracked).
set -o pipefail
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$R
bash scripts/prof_pmc.sh ${R}_dense > gpurun_out/$R/pmc_dense.txt 2>&1 || exit 1
bash scripts/prof_pmc.sh ${R}_sparse --sparse > gpurun_out/$R/pmc_sparse.txt 2>&1 || exit 1
cp gpurun_out/pmc_${R}_dense/traffic.json profiles/pmc_traffic_b256_dense.json
cp gpurun_out/pmc_${R}_sparse/traffic.json profiles/pmc_traffic_b256_sparse.json
# the profiled pass never starts the CPU-baseline worker pool (child processes of a profiled parent are fragile on this
# pool): kernel stats come from the --no-cpu-baseline run under rocprofv3, the bench line with cpu_baseline from a plain run
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python3 bench.py --no-cpu-baseline > gpurun_out/$R/bench_profiled.json 2> gpurun_out/$R/bench_profiled.err || exit 1
python3 bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err || exit 1
python3 bench.py --sparse --no-cpu-baseline > gpurun_out/$R/bench_sparse.json 2> /dev/null || exit 1
cat gpurun_out/$R/bench_default.json
