#!/bin/bash
# scripts/head_modes.sh [rounds] -- the arms of scripts/head_modes.py interleaved on ONE box:
# plain / torchrun child / RCCL before or after the allocations / gloo / shifted allocations / the map first, `rounds` times, then the
# in-process offset sweep, then one arm under rocprofv3 --kernel-trace with a plain arm on either side.
# Output: gpurun_out/head_modes.jsonl (one record per arm), gpurun_out/head_modes.txt (the printed table), gpurun_out/hm_*.err
rounds=${1:-2}
out=gpurun_out
mkdir -p $out
rm -f $out/head_modes.jsonl $out/head_modes.txt
n=0
arm() {  # arm <name> <tag> [launcher...]
  local name=$1 tag=$2; shift 2
  n=$((n + 1))
  local err=$out/hm_$(printf %02d $n)_${name//:/_}_$tag.err
  if [ "$1" = torchrun ]; then
    timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      scripts/head_modes.py --arm $name --tag $tag 2> $err | tee -a $out/head_modes.txt
  else
    timeout -k 10 300 python3 scripts/head_modes.py --arm $name --tag $tag 2> $err | tee -a $out/head_modes.txt
  fi
  local rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "arm $name failed rc=$rc"; tail -5 $err; exit 1; fi
}
for r in $(seq 1 $rounds); do
  arm plain r$r
  arm env_dist r$r torchrun
  arm plain r${r}b
  arm nccl_before r$r
  arm nccl_after r$r
  arm gloo r$r
  arm map_first r$r
  arm dummy:1052672 r$r
  arm dummy:69632 r$r
done
arm offsets sweep
arm plain pre_prof
n=$((n + 1))
( export TMPDIR=/tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/hm_prof -o hm -- python3 scripts/head_modes.py --arm plain --tag rocprof \
    2> $out/hm_$(printf %02d $n)_plain_rocprof.err | tee -a $out/head_modes.txt ) || { echo "rocprof arm failed"; tail -5 $out/hm_$(printf %02d $n)_plain_rocprof.err; exit 1; }
arm plain post_prof
grep -h kpb_alloc $out/hm_*.err | sort | uniq -c | sort -rn | head -40 > $out/head_modes_allocs.txt
echo done
