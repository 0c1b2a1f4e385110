#!/bin/bash
# the two HBM byte-counter passes of tools/gpu_profile_round.sh alone -> gpurun_out/<tag>/traffic.json
TAG=${1:-round}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
W=/tmp/prof_$TAG
mkdir -p $OUT $W
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $W/pmc_fetch -o f -- python $REPO/bench.py --prewarm 0 --warmup 64 --steps 640 --no-cpu-baseline > $W/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $W/pmc_write -o w -- python $REPO/bench.py --prewarm 0 --warmup 64 --steps 640 --no-cpu-baseline > $W/bench_write.log 2>&1
python $REPO/tools/rocpd_summary.py $W --traffic $OUT/traffic.json "RockSample(7,8) 2^20 lanes"
cat $OUT/traffic.json
