#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]  -- kernel-trace stats of a short bench run (GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --lengths uniform --no-self-check "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
rc=$?
find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name '*kernel_trace.csv' -delete
exit $rc
