#!/bin/bash
# PMC passes over a short bench run (run on the GPU box through gpurun):  bash scripts/prof_pmc.sh <tag> [bench.py arguments]
# Each rocprofv3 pass is its own process: counters are collected with --kernel-trace only (never with sys/hip traces).
# Pass 3 holds the matrix-pipe counters: SQ_VALU_MFMA_BUSY_CYCLES (cycles the MFMA ALU is busy, per SIMD: the guide's
# MFMA-utilisation counter) and SQ_INSTS_VALU_MFMA_MOPS_F16 (f16 matrix operations / 512), which is what the split-f16
# kernels issue (the _F32 counter of pass 2 reads 0 for them).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
shift
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants $@"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python scripts/pmc_summary.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
