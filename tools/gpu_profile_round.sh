#!/bin/bash
# One call that produces everything kept under profiles/ for a round (text only; the rocpd databases stay in /tmp).
# usage: tools/gpu_profile_round.sh <tag>      -> gpurun_out/<tag>/{headline.txt,traffic.json,pmc_valu.txt,envs.txt,bench.json}
TAG=${1:-round}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
W=/tmp/prof_$TAG
mkdir -p $OUT $W
cd /tmp
# 1. headline workload: kernel trace, then the two HBM byte counters in separate passes
rocprofv3 --kernel-trace --stats -d $W/trace -o t -- python $REPO/bench.py --steps 2000 --no-cpu-baseline > $W/bench_traced.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $W/pmc_fetch -o f -- python $REPO/bench.py --prewarm 0 --warmup 64 --steps 640 --no-cpu-baseline > $W/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $W/pmc_write -o w -- python $REPO/bench.py --prewarm 0 --warmup 64 --steps 640 --no-cpu-baseline > $W/bench_write.log 2>&1
(echo "# tools/gpu_profile_round.sh $TAG: python bench.py --steps 2000 --no-cpu-baseline under rocprofv3 (MI355X)"; python $REPO/tools/rocpd_summary.py $W | awk '{ if (substr($0,1,1)=="{") print; else print substr($0,1,300) }') > $OUT/headline.txt
python $REPO/tools/rocpd_summary.py $W --traffic $OUT/traffic.json "RockSample(7,8) 2^20 lanes"
# 2. instruction / occupancy counters of the step kernels
cd $REPO
(echo "# rocprofv3 --pmc passes over 'python bench.py --env rock --prewarm 0 --warmup 64 --steps 640 --no-cpu-baseline' (tools/gpu_pmc_env.sh rock), MI355X";
 echo "# divide SQ_INSTS_* by SQ_WAVES for per-wave counts (128 lanes at two lanes per thread, 256 at four); steps_kernel launches are 64 steps each";
 bash tools/gpu_pmc_env.sh rock 2>/dev/null) > $OUT/pmc_valu.txt
# 3. every env, the fused rollouts and the heuristic policy: kernel traces
cd /tmp
: > $OUT/envs.txt
for e in rock15 tag battleship tiger network; do
  rm -rf $W/e; mkdir -p $W/e
  rocprofv3 --kernel-trace --stats -d $W/e/trace -o e -- python $REPO/bench.py --env $e --steps 1500 --warmup 300 --no-cpu-baseline > $W/e/bench_traced.log 2>&1
  (echo "##### bench.py --env $e --steps 1500 --warmup 300 --no-cpu-baseline"; python $REPO/tools/rocpd_summary.py $W/e | awk '{ if (substr($0,1,1)=="{") print; else print substr($0,1,260) }') >> $OUT/envs.txt
done
for e in rock15 rock; do
  rm -rf $W/e; mkdir -p $W/e
  rocprofv3 --kernel-trace --stats -d $W/e/trace -o e -- python $REPO/bench.py --env $e --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100 > $W/e/bench_traced.log 2>&1
  (echo "##### bench.py --env $e --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100"; python $REPO/tools/rocpd_summary.py $W/e | awk '{ if (substr($0,1,1)=="{") print; else print substr($0,1,260) }') >> $OUT/envs.txt
done
for e in rock rock15 tag; do
  rm -rf $W/e; mkdir -p $W/e
  rocprofv3 --kernel-trace --stats -d $W/e/trace -o e -- python $REPO/bench.py --env $e --mode heuristic --steps 1024 --warmup 128 > $W/e/bench_traced.log 2>&1
  (echo "##### bench.py --env $e --mode heuristic --steps 1024 --warmup 128"; python $REPO/tools/rocpd_summary.py $W/e | awk '{ if (substr($0,1,1)=="{") print; else print substr($0,1,260) }') >> $OUT/envs.txt
done
# 4. the default bench line, unprofiled
cd $REPO
timeout 600 python bench.py 2>/dev/null | tail -1 > $OUT/bench.json
ls -la $OUT
