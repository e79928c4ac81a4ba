#!/bin/bash
# GPU box: the other configurations of DESIGN 5's table, one short bench run each (no CPU baseline, no ragged / split legs)
mkdir -p gpurun_out/r3
run() { tag=$1; shift; timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --lengths uniform --no-self-check "$@" > gpurun_out/r3/bench_r3_$tag.json 2> gpurun_out/r3/bench_r3_$tag.err || exit 1; }
run batch16 --batch 16 && run batch32 --batch 32 && run configs1 --batch 1 --views 1 && run configs3 --batch 1 --prompt temporal && run symbolic --prompt symbolic && echo other_configs_done
