#!/bin/bash
# Is the step time of the keypoint-only pipeline quantised (r05 and r06 read 9.000 ms per 256 pairs on three different boxes)?  Vary the batch by a few
# pairs: a continuous time follows the work, a quantised one sticks.  KPB_WAIT_MODE 0 = parked wait, 1 = spin.
for m in 0 1; do for b in 244 248 252 256 260 264; do
KPB_WAIT_MODE=$m python bench.py --no-cpu-baseline --no-variants --distinct 32 --steps 100 --sparse --pairs-per-step $b > gpurun_out/q.json 2> gpurun_out/q.err || { tail -3 gpurun_out/q.err; exit 1; }
python - "$m $b" gpurun_out/q.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = r["roofline"]["kernels_ms_per_step"]
print("mode/batch", sys.argv[1], "value %.0f ms/step %.4f per pair us %.3f sum kernels %.3f" % (r["value"], r["ms_per_step"], 1e3 * r["ms_per_step"] / r["config"]["pairs_per_step_per_gpu"], sum(k.values())), flush=True)
PY
done; done
