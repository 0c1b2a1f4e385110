#!/bin/bash
# One call that produces everything kept under profiles/ for a round (text only; the rocpd databases stay in /tmp).
# usage: tools/gpu_profile_round.sh <tag>  -> gpurun_out/<tag>/{headline_steps20.txt,headline.txt,envs.txt,bench*.json(l),
#        layout_probe_*.txt,small_shards_*.txt,sinks*.txt,single_step.txt,valu_microbench.json,pmc_valu.json,pmc_valu.txt,isa_mix.json}
TAG=${1:-round}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
W=/tmp/prof_$TAG
mkdir -p $OUT $W
trace() {   # <out file> <header> bench args...: kernel trace of one bench command, summarised
  local out=$1 hdr=$2; shift 2
  rm -rf $W/t; mkdir -p $W/t
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $W/t/trace -o t -- python $REPO/bench.py "$@" > $W/t/bench_traced.log 2>&1)
  (echo "##### python bench.py $* under rocprofv3 --kernel-trace --stats (MI355X)$hdr"; python $REPO/tools/rocpd_summary.py $W/t | awk '{ if (substr($0,1,1)=="{") print; else print substr($0,1,280) }'; tail -1 $W/t/bench_traced.log) >> $out
}
# (the whole GPU suite first: the profiles belong to a tree whose parity is green)
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest_gpu.log
# 0. the bench exactly as the driver runs it (one 20-step launch per timed region), then the default (256-step launches)
: > $OUT/headline_steps20.txt; trace $OUT/headline_steps20.txt "" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
: > $OUT/headline.txt; trace $OUT/headline.txt "" --steps 2000 --no-cpu-baseline --no-extras
for l in columns blocked narrow returns; do trace $OUT/headline.txt " — layout $l" --steps 2000 --layout $l --no-cpu-baseline --no-extras; done
# 1. every env (packed layout), the fused rollouts and the heuristic policy: kernel traces
: > $OUT/envs.txt
for e in rock15 stochrock tag battleship battleship5 tiger network; do trace $OUT/envs.txt "" --env $e --steps 1500 --warmup 300 --no-cpu-baseline --no-extras; done
for e in rock15 rock; do trace $OUT/envs.txt "" --env $e --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100; done
for e in rock rock15 tag; do trace $OUT/envs.txt "" --env $e --mode heuristic --steps 1024 --warmup 128; done
# 2. the counters first (step 4 below used to come last): the bench lines then read THIS tree's counters from profiles/
cd $REPO
(cd tools && hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_microbench valu_microbench.hip 2>/dev/null); /tmp/valu_microbench > $OUT/valu_microbench.json 2>/dev/null
python tools/isa_mix.py --costs $OUT/valu_microbench.json --json $OUT/isa_mix.json > $OUT/isa_mix.txt 2>/dev/null
bash tools/gpu_pmc_valu.sh ${TAG}_valu "planners headline envs shards" > $OUT/pmc_valu.log 2>&1
cp $REPO/gpurun_out/${TAG}_valu/pmc_valu.json $REPO/gpurun_out/${TAG}_valu/pmc_valu.txt $OUT/ 2>/dev/null
cp $OUT/pmc_valu.json profiles/${TAG}_pmc_valu.json; cp $OUT/isa_mix.json profiles/${TAG}_isa_mix.json      # (on the box's copy of the tree)
# 2b. the bench lines, unprofiled: default, as the driver runs it, two ranks on the one GPU, the other envs and layouts
timeout 600 python bench.py 2>/dev/null | tail -1 > $OUT/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_steps20.json
timeout 600 python bench.py --gpus 2 --share-gpus --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_gpus2.json
: > $OUT/bench_envs.jsonl
for e in rock15 stochrock tag battleship battleship5 tiger network; do
  for l in packed columns; do timeout 600 python bench.py --env $e --layout $l --no-cpu-baseline --no-extras 2>/dev/null | tail -1 >> $OUT/bench_envs.jsonl; done
done
: > $OUT/bench_modes.jsonl
timeout 600 python bench.py --env rock15 --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100 2>/dev/null | tail -1 >> $OUT/bench_modes.jsonl
timeout 600 python bench.py --env rock --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100 2>/dev/null | tail -1 >> $OUT/bench_modes.jsonl
for e in rock rock15 tag; do timeout 600 python bench.py --env $e --mode heuristic --steps 1024 --warmup 128 2>/dev/null | tail -1 >> $OUT/bench_modes.jsonl; done
# 3. placement / layout probe, small shards (packed and columns), one step per launch
PP_LAYOUTS=columns,blocked,packed,narrow PP_ALLOCS=6 timeout 600 python tools/gpu_layout_probe.py 256 rock > $OUT/layout_probe_256.txt 2>/dev/null
PP_LAYOUTS=columns,blocked,packed,narrow timeout 600 python tools/gpu_layout_probe.py 20 rock > $OUT/layout_probe_20.txt 2>/dev/null
SINK_K=64,256 timeout 900 python tools/gpu_sinks_probe.py 2>/dev/null > $OUT/sinks.txt
SINK_ENVS=rock,tag,battleship,network SINK_SINKS=packed,returns SINK_SIZES=131072,262144,524288 SINK_K=64,256 timeout 900 python tools/gpu_sinks_probe.py 2>/dev/null > $OUT/sinks_shards.txt
(rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCP_PENDING[A-Za-z0-9_]*" | sort -u | tr '\n' ' ') > $OUT/avail_store_counters.txt
timeout 900 python tools/gpu_small_shards.py - 2>/dev/null > $OUT/small_shards_packed.txt
SHARD_LAYOUT=columns SHARD_ENVS=rock,tag,battleship timeout 900 python tools/gpu_small_shards.py - 2>/dev/null > $OUT/small_shards_columns.txt
timeout 600 python tools/gpu_single_step_probe.py - 20 2>/dev/null > $OUT/single_step.txt
ls -la $OUT
