#!/bin/bash
# usage: tools/prof_env.sh <tag> -- like prof.sh; environment variables set by the caller reach the bench (export them first)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --new-tokens 4 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
rc=$?
find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name '*kernel_trace.csv' -delete
exit $rc
