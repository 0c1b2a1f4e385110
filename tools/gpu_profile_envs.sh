#!/bin/bash
# rocprofv3 kernel-trace of bench.py for every env (and the rollout mode); summaries -> gpurun_out/<tag>/
TAG=${1:-envs}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
for e in rock rock15 tag battleship tiger network roll; do mkdir -p /tmp/prof_$e; done
for e in rock rock15 tag battleship tiger network; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$e/trace -o $e -- python $REPO/bench.py --env $e --steps 1500 --warmup 600 --no-cpu-baseline > /tmp/prof_$e/bench_traced.log 2>&1
  (echo "##### bench.py --env $e --steps 1500 --warmup 600"; python $REPO/tools/rocpd_summary.py /tmp/prof_$e | cut -c1-260) >> $OUT/summary.txt
done
rocprofv3 --kernel-trace --stats -d /tmp/prof_roll/trace -o roll -- python $REPO/bench.py --env rock15 --mode rollout --lanes-per-gpu 2097152 --steps 20 --warmup 2 > /tmp/prof_roll/bench_traced.log 2>&1
(echo "##### bench.py --env rock15 --mode rollout --lanes-per-gpu 2097152 --steps 20"; python $REPO/tools/rocpd_summary.py /tmp/prof_roll | cut -c1-260) >> $OUT/summary.txt
cd $REPO
wc -l $OUT/summary.txt
