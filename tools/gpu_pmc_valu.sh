#!/bin/bash
# VALU-issue counters of the kernels whose bound is instruction issue ("report int-op rate"): the fused
# rollout (configs[4]), the heuristic-policy loop, and the fused step launches in both timed shapes (64 and 20 steps per
# launch), plus the HBM byte counters of the 20-step shape.  One rocprofv3 --pmc pass per counter group (never combined
# with other trace domains), summaries into gpurun_out/<tag>/.
# usage: tools/gpu_pmc_valu.sh <tag>     -> gpurun_out/<tag>/{pmc_valu.json, pmc_valu.txt}
TAG=${1:-pmcv}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
W=/tmp/pmcv_$TAG
rm -rf $W; mkdir -p $OUT $W
cd /tmp
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
run() {   # name, bench args...
  local name=$1; shift
  mkdir -p $W/$name
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 -d $W/$name/p1 -o p1 -- python $REPO/bench.py "$@" > $W/$name/p1.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 -d $W/$name/p2 -o p2 -- python $REPO/bench.py "$@" > $W/$name/p2.log 2>&1
  echo "$name: $*" >> $W/commands.txt
}
run rollout_rock15 --env rock15 --mode rollout --lanes-per-gpu 2097152 --steps 20 --warmup 10
run rollout_rock --env rock --mode rollout --lanes-per-gpu 2097152 --steps 20 --warmup 10
run rollout_tag --env tag --mode rollout --lanes-per-gpu 2097152 --steps 20 --warmup 10
run heuristic_rock --env rock --mode heuristic --prewarm 0 --warmup 128 --steps 1024
run heuristic_rock15 --env rock15 --mode heuristic --prewarm 0 --warmup 128 --steps 1024
run heuristic_tag --env tag --mode heuristic --prewarm 0 --warmup 128 --steps 1024
for e in rock rock15 tag tiger network battleship; do
  run step64_$e --env $e --prewarm 0 --warmup 64 --steps 640 --seeds 0 --repeats 1 --no-cpu-baseline
done
run step20_rock --env rock --gpus 1 --steps 20 --warmup 5 --prewarm 0 --seeds 0 --no-cpu-baseline
# HBM bytes of the driver's 20-step launch: FETCH_SIZE and WRITE_SIZE in separate passes
mkdir -p $W/step20_rock/pmc_fetch $W/step20_rock/pmc_write
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $W/step20_rock/pmc_fetch -o f -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --prewarm 0 --seeds 0 --no-cpu-baseline > $W/step20_rock/f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $W/step20_rock/pmc_write -o w -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --prewarm 0 --seeds 0 --no-cpu-baseline > $W/step20_rock/w.log 2>&1
cd $REPO
python tools/pmc_valu_summary.py $W $OUT
ls -la $OUT
