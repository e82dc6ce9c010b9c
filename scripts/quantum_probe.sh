#!/bin/bash
# Step time against the batch size (r06): does the time per pair depend on how long a step is?  scripts/quantum_probe.sh [--sparse] -> microseconds per pair
# for batches of 64 .. 264 pairs (bench.py --pairs-per-step; everything else as the headline run).  Record: profiles/r06_step_time_vs_batch*.txt
for b in ${BATCHES:-64 96 128 160 192 224 244 248 252 256 260 264}; do
python bench.py --no-cpu-baseline --no-variants --distinct 32 --steps ${STEPS:-100} --pairs-per-step $b "$@" > gpurun_out/q.json 2> gpurun_out/q.err || { tail -3 gpurun_out/q.err; exit 1; }
python - "$b" gpurun_out/q.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = r["roofline"]["kernels_ms_per_step"]
print("batch", sys.argv[1], "value %.0f ms/step %.4f per pair us %.3f sum kernels %.3f dominant %.3f" % (r["value"], r["ms_per_step"], 1e3 * r["ms_per_step"] / r["config"]["pairs_per_step_per_gpu"], sum(k.values()), r["roofline"]["avg_ms"]), flush=True)
PY
done
