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
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err || exit 1
python bench.py --sparse --no-cpu-baseline > gpurun_out/$R/bench_sparse.json 2> /dev/null || exit 1
cat gpurun_out/$R/bench_default.json
