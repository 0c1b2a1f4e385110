#!/bin/bash
# VALU-issue counters of the kernels whose bound is instruction issue ("report int-op rate"): the fused
# rollout (configs[4]), the heuristic-policy loop, and the fused step launches in both timed shapes (256 and 20 steps per
# launch) and every sink, plus the HBM byte counters of the headline's shapes.  One rocprofv3 --pmc pass per
# counter group (never combined with other trace domains), summaries into gpurun_out/<tag>/.
# usage: tools/gpu_pmc_valu.sh <tag> [sets]    -> gpurun_out/<tag>/{pmc_valu.json, pmc_valu.txt}
#   sets (default "planners headline envs"): planners = rollouts + heuristic loops; headline = RockSample(7,8) in every sink
#   (packed, columns, blocked, narrow, returns) x (256, 20) steps per launch + HBM byte passes; envs = the other envs' fused
#   launches (packed and columns); shards = RockSample(7,8) at 2^17 / 2^18 / 2^19 lanes.
#   Workloads that are not re-recorded are carried over from the newest profiles/*_pmc_valu.json, each with the source hash
#   it was recorded under (bench.py: counters_stale).
TAG=${1:-pmcv}
SETS=${2:-"planners headline envs"}
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
SQ3="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum"
SQ4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum"
stalls() {   # name, bench args...: what the store path waits for (L2 write requests towards the fabric and their stalls, the
             # vector cache's pending stalls) — two more passes, kept apart so that a counter this build of rocprofv3 does not
             # know costs one pass, not all
  local name=$1; shift
  mkdir -p $W/$name
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ3 -d $W/$name/p3 -o p3 -- python $REPO/bench.py "$@" > $W/$name/p3.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ4 -d $W/$name/p4 -o p4 -- python $REPO/bench.py "$@" > $W/$name/p4.log 2>&1
}
traffic() {   # name, bench args...: FETCH_SIZE and WRITE_SIZE in separate passes
  local name=$1; shift
  mkdir -p $W/$name/pmc_fetch $W/$name/pmc_write
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $W/$name/pmc_fetch -o f -- python $REPO/bench.py "$@" > $W/$name/f.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $W/$name/pmc_write -o w -- python $REPO/bench.py "$@" > $W/$name/w.log 2>&1
}
sfx() { if [ "$1" = columns ]; then echo ""; else echo "_$1"; fi; }
S256="--prewarm 0 --warmup 256 --steps 1024 --seeds 0 --repeats 1 --no-cpu-baseline --no-extras"
S20="--gpus 1 --steps 20 --warmup 5 --prewarm 0 --seeds 0 --no-cpu-baseline --no-extras"
for set in $SETS; do
  case $set in
  planners)
    run rollout_rock15 --env rock15 --mode rollout --lanes-per-gpu 2097152 --steps 20 --warmup 10
    run rollout_rock --env rock --mode rollout --lanes-per-gpu 2097152 --steps 20 --warmup 10
    run rollout_tag --env tag --mode rollout --lanes-per-gpu 2097152 --steps 20 --warmup 10
    run heuristic_rock --env rock --mode heuristic --prewarm 0 --warmup 128 --steps 1024
    run heuristic_rock15 --env rock15 --mode heuristic --prewarm 0 --warmup 128 --steps 1024
    run heuristic_tag --env tag --mode heuristic --prewarm 0 --warmup 128 --steps 1024
    # what the heuristic loop's CHECKs move: the side statistics are seven [K][N] arrays read-modify-written per CHECK
    traffic heuristic_rock15 --env rock15 --mode heuristic --prewarm 0 --warmup 128 --steps 1024
    traffic heuristic_rock --env rock --mode heuristic --prewarm 0 --warmup 128 --steps 1024 ;;
  headline)
    for l in packed columns blocked narrow returns; do
      k=256; if [ $l = columns ] || [ $l = blocked ]; then k=64; fi      # pomdp_fuse_steps: RockSample's 13-byte layouts stay at 64
      run step${k}_rock$(sfx $l) --env rock --layout $l $S256
      run step20_rock$(sfx $l) --env rock --layout $l $S20
      traffic step20_rock$(sfx $l) --env rock --layout $l $S20
    done
    traffic step256_rock_packed --env rock --layout packed $S256
    traffic step64_rock --env rock --layout columns $S256
    stalls step256_rock_packed --env rock --layout packed $S256; stalls step256_rock_narrow --env rock --layout narrow $S256
    stalls step64_rock --env rock --layout columns $S256; stalls step64_rock_blocked --env rock --layout blocked $S256 ;;
  sinks)    # the headline workload's sinks at 256 steps per launch only (a quick A/B of what a sink costs)
    for l in packed narrow returns; do run step256_rock$(sfx $l) --env rock --layout $l $S256; done ;;
  envs)
    for e in rock15 tag tiger network battleship battleship5; do
      run step256_${e}_packed --env $e --layout packed $S256
      run step20_${e}_packed --env $e --layout packed $S20
    done
    run step64_tag --env tag --layout columns $S256
    for e in network battleship; do run step256_$e --env $e --layout columns $S256; done
    # BASELINE.json configs[3] per GPU: BattleShip 10x10 at 2^19 lanes (bench.py: configs.battleship)
    run step256_battleship_packed_2e19 --env battleship --layout packed --lanes-per-gpu 524288 $S256
    run step20_battleship_packed_2e19 --env battleship --layout packed --lanes-per-gpu 524288 $S20 ;;
  shards)   # the shards a 2^20-lane batch leaves per GPU at 8 / 4 / 2 GPUs (strong scaling: DESIGN.md §7)
    for lg in 17 18 19; do       # both launch shapes: bench.py's strong_scaling.frac_of_floor looks the shard up by size and length
      run step256_rock_packed_2e$lg --env rock --layout packed --lanes-per-gpu $((1 << lg)) $S256
      run step20_rock_packed_2e$lg --env rock --layout packed --lanes-per-gpu $((1 << lg)) $S20
    done ;;
  esac
done
cd $REPO
python tools/pmc_valu_summary.py $W $OUT "$(ls -t profiles/*_pmc_valu.json 2>/dev/null | head -1)"
ls -la $OUT
