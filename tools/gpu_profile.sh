#!/bin/bash
# rocprofv3 kernel trace + (separate passes) HBM byte counters for bench.py's workload.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -x
TAG=${1:-prof}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python $OLDPWD/bench.py --steps 500 --no-cpu-baseline "$@" > $OUT/bench_traced.log 2>&1
tail -2 $OUT/bench_traced.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o $TAG -- python $OLDPWD/bench.py --steps 100 --no-cpu-baseline "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o $TAG -- python $OLDPWD/bench.py --steps 100 --no-cpu-baseline "$@" > $OUT/bench_write.log 2>&1
cd $OLDPWD
find $OUT -type f | head -40
ls -la $OUT/trace/* | head
