#!/bin/bash
# usage: tools/pmc_bin.sh <tag> <binary> <arg> -- <counters...>   one PMC pass of a micro-benchmark binary (GPU box)
tag=$1; bin=$2; arg=$3; shift 4
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o pmc -- $GRAFT_REPO_ROOT/$bin $arg > $out.log 2>&1
rc=$?
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out > $out.summary.txt 2>&1
find $out -name '*.csv' -size +2000k -delete
exit $rc
