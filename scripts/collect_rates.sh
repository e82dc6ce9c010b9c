#!/bin/bash
# Task rates, parity sweep and latency records of a round, run on the GPU box through gpurun:  bash scripts/collect_rates.sh r05
# (the companion of collect_evidence.sh / collect_other_nets.sh: everything here runs on the SAME library; every record carries its hash)
R=${1:-r05}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$R
LIB=$(python3 -c "import hashlib;print(hashlib.sha256(open('keypoint_bench_amd/libkpb.so','rb').read()).hexdigest()[:12])")
echo "library $LIB" | tee gpurun_out/$R/rates_build.txt
run() { out=$1; shift; timeout -k 10 400 python3 scripts/runner_rate.py "$@" > gpurun_out/$R/$out.json 2> gpurun_out/$R/$out.err && python3 - gpurun_out/$R/$out.json $LIB <<'PY' || echo "$out failed"
import json, sys
d = json.load(open(sys.argv[1])); d["lib_sha256"] = sys.argv[2]
json.dump(d, open(sys.argv[1], "w"), indent=1)
print(sys.argv[1].split("/")[-1], {k: (v.get("pairs_per_s") if isinstance(v, dict) else v) for k, v in d.items() if k not in ("items", "descriptors")})
PY
}
run runner_rate
run runner_rate_host --host
run runner_rate_u8 --host --u8
run runner_rate_seq --sequence
run runner_rate_seq_xfeat --sequence --model XFeat
run runner_rate_files_png --files png --pairs 512
run runner_rate_files_jpeg --files jpeg --pairs 512
run runner_rate_files_ppm --files ppm --pairs 2048
timeout -k 10 600 python3 scripts/parity_sweep.py 256 gpurun_out/$R/parity_sweep_256.json --also-fp32 > gpurun_out/$R/parity_sweep_256.txt 2>&1 || echo "parity sweep failed"
tail -3 gpurun_out/$R/parity_sweep_256.txt
timeout -k 10 300 python3 bench.py --steps 500 --no-cpu-baseline --no-variants > gpurun_out/$R/bench_500_steps.json 2> gpurun_out/$R/bench_500_steps.err || echo "500-step run failed"
