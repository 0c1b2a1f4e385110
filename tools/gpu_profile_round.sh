#!/bin/bash
# One call that produces everything kept under profiles/ for a round (text only; the rocpd databases stay in /tmp).
# usage: tools/gpu_profile_round.sh <tag>      -> gpurun_out/<tag>/{headline.txt,headline_steps20.txt,traffic.json,pmc_rock.txt,pmc_envs.txt,envs.txt,bench*.json(l),small_shards.txt,single_step.txt,valu_microbench.json,pmc_valu.json,pmc_valu.txt}
TAG=${1:-round}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
W=/tmp/prof_$TAG
mkdir -p $OUT $W
cd /tmp
# 0. the bench exactly as the driver runs it (one 20-step launch per timed region), under the kernel trace
mkdir -p $W/drv
rocprofv3 --kernel-trace --stats -d $W/drv/trace -o t -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $W/drv/bench_traced.log 2>&1
(echo "# tools/gpu_profile_round.sh $TAG: python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline under rocprofv3 --kernel-trace --stats (MI355X)"; echo "# the timed regions are the 20-step launches of steps_quad_kernel (warm-up: 5-step launches of steps_kernel<., 4, true>); the bench line printed under the tracer follows"; python $REPO/tools/rocpd_summary.py $W/drv | awk '{ if (substr($0,1,1)=="{") print; else print substr($0,1,300) }'; tail -1 $W/drv/bench_traced.log) > $OUT/headline_steps20.txt
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
 bash tools/gpu_pmc_env.sh rock 2>/dev/null) > $OUT/pmc_rock.txt
# 2b. instructions per wave-step of every env's fused launch (one PMC pass each)
: > $OUT/pmc_envs.txt
for e in rock rock15 tag battleship battleship5 tiger network; do
  (echo "##### tools/gpu_pmc_quick.sh $e"; bash tools/gpu_pmc_quick.sh $e 2>/dev/null) >> $OUT/pmc_envs.txt
done
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
# 4. the bench lines, unprofiled: default, as the driver runs it, two ranks on the one GPU, the other envs
cd $REPO
timeout 600 python bench.py 2>/dev/null | tail -1 > $OUT/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_steps20.json
timeout 600 python bench.py --gpus 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_gpus2.json
: > $OUT/bench_envs.jsonl
for e in rock15 stochrock tag battleship battleship5 tiger network; do timeout 600 python bench.py --env $e --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_envs.jsonl; done
: > $OUT/bench_modes.jsonl
timeout 600 python bench.py --env rock15 --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100 2>/dev/null | tail -1 >> $OUT/bench_modes.jsonl
timeout 600 python bench.py --env rock --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100 2>/dev/null | tail -1 >> $OUT/bench_modes.jsonl
for e in rock rock15 tag; do timeout 600 python bench.py --env $e --mode heuristic --steps 1024 --warmup 128 2>/dev/null | tail -1 >> $OUT/bench_modes.jsonl; done
timeout 600 python tools/gpu_small_shards.py - 2>/dev/null > $OUT/small_shards.txt
timeout 600 python tools/gpu_single_step_probe.py - 20 2>/dev/null > $OUT/single_step.txt
# 5. the VALU-issue counters of the compute-bound kernels and of both timed launch shapes (tools/gpu_pmc_valu.sh), and the
#    instruction costs they are priced with
tools/valu_microbench > $OUT/valu_microbench.json 2>/dev/null
bash tools/gpu_pmc_valu.sh ${TAG}_valu > $OUT/pmc_valu.log 2>&1
cp $REPO/gpurun_out/${TAG}_valu/pmc_valu.json $REPO/gpurun_out/${TAG}_valu/pmc_valu.txt $OUT/ 2>/dev/null
ls -la $OUT
