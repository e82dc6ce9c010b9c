#!/bin/bash
# scripts/head_modes2.sh -- second step of the head's time modes: does the mode belong to the ALLOCATION?
# Three processes in a row, each: the pipeline's own map + 4 more 40 GB maps side by side, the step re-timed on each buffer round robin.
out=gpurun_out
mkdir -p $out
rm -f $out/head_modes2.jsonl $out/head_modes2.txt
for r in 1 2 3; do
  timeout -k 10 400 python3 scripts/head_modes.py --arm candidates:4 --tag r$r --out $out/head_modes2.jsonl 2> $out/hm2_$r.err | tee -a $out/head_modes2.txt || { tail -5 $out/hm2_$r.err; exit 1; }
  grep "^round" $out/hm2_$r.err | tee -a $out/head_modes2.txt
done
echo done
