#!/bin/bash
# Kernel-variant experiments on the GPU box (through gpurun):  bash scripts/variants.sh OUTFILE "ENV1=a ENV2=b" "ENV1=c" ...
# Each variant is one bench.py process (the knobs are read once per process); prints ms_per_step and the per-kernel
# HIP-event times of bench.py's roofline leg.
OUT=$1; shift
mkdir -p gpurun_out
: > gpurun_out/$OUT
for v in "$@"; do
  echo "== $v" >> gpurun_out/$OUT
  env $v python bench.py --no-cpu-baseline --steps 8 --warmup 2 --distinct 16 ${BENCH_ARGS} 2>> gpurun_out/$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'variant', (d.get('variant') or {}).get('ms_per_step'))
k=d['roofline']['kernels_ms_per_step']
print(' '.join('%s=%.3f'%(a,b) for a,b in list(k.items())[:14]))
" >> gpurun_out/$OUT
done
cat gpurun_out/$OUT
