timeout 900 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/r03g_pytest.log 2>&1; tail -8 gpurun_out/r03g_pytest.log
SHARD_ENVS=battleship python tools/gpu_small_shards.py - > gpurun_out/r03g_small_shards.txt 2>&1
rm -f gpurun_out/r03g_bench_bs.jsonl
for e in battleship battleship5; do timeout 300 python bench.py --env $e --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r03g_bench_bs.jsonl; done
for e in battleship battleship5; do echo "##### $e"; bash tools/gpu_pmc_quick.sh $e 2>/dev/null; done > gpurun_out/r03g_pmc_bs.txt
