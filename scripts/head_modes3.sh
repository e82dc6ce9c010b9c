#!/bin/bash
# scripts/head_modes3.sh -- third step: with the descriptor map PLACED by measurement (PairPipeline._place_map) the launch
# paths must agree: plain / torchrun child / under rocprofv3, interleaved on one box, then bench.py itself plain and --spawn, then bench.py under
# rocprofv3 --kernel-trace --stats (whose per-kernel average must match the line's HIP-event average).
out=gpurun_out
mkdir -p $out
rm -f $out/head_modes3.jsonl $out/head_modes3.txt
n=0
arm() {
  local name=$1 tag=$2; shift 2
  n=$((n + 1))
  local err=$out/hm3_$(printf %02d $n)_${name}_$tag.err
  if [ "$1" = torchrun ]; then
    timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29700 + n)) \
      scripts/head_modes.py --place --arm $name --tag $tag --out $out/head_modes3.jsonl 2> $err | tee -a $out/head_modes3.txt
  else
    timeout -k 10 300 python3 scripts/head_modes.py --place --arm $name --tag $tag --out $out/head_modes3.jsonl 2> $err | tee -a $out/head_modes3.txt
  fi
  local rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "arm $name failed rc=$rc"; tail -5 $err; exit 1; fi
}
for r in 1 2 3; do
  arm plain r$r
  arm env_dist r$r torchrun
done
n=$((n + 1))
( export TMPDIR=/tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/hm3_prof -o hm -- python3 scripts/head_modes.py --place --arm plain --tag rocprof --out $out/head_modes3.jsonl \
    2> $out/hm3_$(printf %02d $n)_plain_rocprof.err | tee -a $out/head_modes3.txt ) || { echo "rocprof arm failed"; tail -5 $out/hm3_$(printf %02d $n)_plain_rocprof.err; exit 1; }
arm plain post_prof
for r in 1 2; do
  python3 bench.py --no-cpu-baseline --no-variants > $out/hm3_bench_plain_$r.json 2> $out/hm3_bench_plain_$r.err || { echo plain bench failed; tail -5 $out/hm3_bench_plain_$r.err; exit 1; }
  python3 bench.py --spawn --no-cpu-baseline --no-variants > $out/hm3_bench_spawn_$r.json 2> $out/hm3_bench_spawn_$r.err || { echo spawn bench failed; tail -5 $out/hm3_bench_spawn_$r.err; exit 1; }
done
( export TMPDIR=/tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/hm3_bench_prof -- python3 bench.py --no-cpu-baseline --no-variants > $out/hm3_bench_profiled.json 2> $out/hm3_bench_profiled.err ) || { echo profiled bench failed; tail -5 $out/hm3_bench_profiled.err; exit 1; }
python3 - <<'PY' | tee -a gpurun_out/head_modes3.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/hm3_bench_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    k = r["roofline"]
    print("%-44s value %8.1f sustained %s ms/step %.3f head %.3f placement %s" % (f.split("/")[-1], r["value"], r.get("value_sustained"), r["ms_per_step"], k["kernels_ms_per_step"].get("alike_head_dense", 0), r["config"].get("placement")))
PY
find $out/hm3_bench_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -4 {}' | tee -a $out/head_modes3.txt
echo done
