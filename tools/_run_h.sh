timeout 900 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/r03h_pytest.log 2>&1; tail -8 gpurun_out/r03h_pytest.log
python tools/gpu_single_step_probe.py - 20 > gpurun_out/r03h_single_step.txt 2>&1
bash tools/gpu_pmc_valu.sh r03h > gpurun_out/r03h_pmc.log 2>&1
python tools/gpu_small_shards.py - > gpurun_out/r03h_small_shards.txt 2>&1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r03h_bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03h_bench_steps20.json
timeout 600 python bench.py --gpus 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03h_bench_gpus2.json
