#!/bin/bash
# round 2, first GPU call: new timed-kernel tests, whole suite, bench as the driver runs it, sync-latency knob, small shards
mkdir -p gpurun_out/r02a
cd /root/repo
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_timed_kernels.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02a/tests_new.log
( timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_timed_kernels.py 2>&1 | tail -15 ) > gpurun_out/r02a/tests_all.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a/bench_20.json 2> gpurun_out/r02a/bench_20.err
python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
for w in 0 50 1000; do
  echo "ROC_ACTIVE_WAIT_TIMEOUT=$w" >> gpurun_out/r02a/sync.log
  ROC_ACTIVE_WAIT_TIMEOUT=$w timeout 120 python tools/sync_probe.py >> gpurun_out/r02a/sync.log 2>&1
  ROC_ACTIVE_WAIT_TIMEOUT=$w python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --seeds 0 > gpurun_out/r02a/bench_20_wait$w.json 2>/dev/null
done
timeout 900 python tools/gpu_small_shards.py - gym_pomdp_amd/_lib/libpomdp_hip_q12.so > gpurun_out/r02a/small_shards.log 2>&1
