#!/bin/bash
# Round evidence, run on the GPU box through gpurun:  bash scripts/collect_evidence.sh r03
# Order matters: the PMC passes come first and their per-kernel HBM traffic is put where bench.py looks for it
# (profiles/pmc_traffic_*.json), so the bench line recorded afterwards carries the traffic of THIS build.
# Leaves everything under gpurun_out/$1/; scripts/publish_evidence.py copies the summaries into profiles/ (tracked).
set -o pipefail
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$R
bash scripts/prof_pmc.sh ${R}_dense > gpurun_out/$R/pmc_dense.txt 2>&1 || exit 1
bash scripts/prof_pmc.sh ${R}_sparse --sparse > gpurun_out/$R/pmc_sparse.txt 2>&1 || exit 1
cp gpurun_out/pmc_${R}_dense/traffic.json profiles/pmc_traffic_b256_dense.json
cp gpurun_out/pmc_${R}_sparse/traffic.json profiles/pmc_traffic_b256_sparse.json
# the profiled pass starts no child process (no CPU-baseline pool, no strict-fp32 companion): kernel stats come from this run
# under rocprofv3, the bench line with cpu_baseline / variant_fp32 from the plain run that follows
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python3 bench.py --no-cpu-baseline --no-variants > gpurun_out/$R/bench_profiled.json 2> gpurun_out/$R/bench_profiled.err || exit 1
python3 bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err || exit 1
python3 bench.py --sparse --no-cpu-baseline --no-variants > gpurun_out/$R/bench_sparse.json 2> /dev/null || exit 1
# the multi-GPU launch path at world = 1: a child torchrun, one rank over RCCL
python3 bench.py --spawn --no-cpu-baseline --no-variants > gpurun_out/$R/bench_spawn_w1.json 2> gpurun_out/$R/bench_spawn_w1.err || echo "spawn run failed"
# the path main.py + install() really takes: one pair at a time through the drop-ins
python3 scripts/single_pair_latency.py > gpurun_out/$R/single_pair_latency.txt 2>&1 || echo "latency script failed"
python3 scripts/single_pair_kernels.py >> gpurun_out/$R/single_pair_latency.txt 2>&1 || echo "per-kernel latency script failed"
python3 scripts/isa_lint.py > gpurun_out/$R/isa_lint.txt 2>&1 || echo "ISA LINT FAILED"
cat gpurun_out/$R/bench_default.json
