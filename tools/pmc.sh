#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   -- one PMC pass (own run, no tracing besides the kernel trace) of a short bench
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --new-tokens 12 --no-cpu-baseline --lengths uniform --no-self-check > $out.log 2>&1
rc=$?
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out > $out.summary.txt 2>&1
find $out -name '*.csv' -size +2000k -delete
exit $rc
