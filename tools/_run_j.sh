bash tools/gpu_pmc_valu.sh r03j > gpurun_out/r03j_pmc.log 2>&1
: > gpurun_out/r03j_bench_modes.jsonl
timeout 600 python bench.py --env rock15 --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100 2>/dev/null | tail -1 >> gpurun_out/r03j_bench_modes.jsonl
timeout 600 python bench.py --env rock --mode rollout --lanes-per-gpu 2097152 --steps 200 --warmup 100 2>/dev/null | tail -1 >> gpurun_out/r03j_bench_modes.jsonl
for e in rock rock15 tag; do timeout 600 python bench.py --env $e --mode heuristic --steps 1024 --warmup 128 2>/dev/null | tail -1 >> gpurun_out/r03j_bench_modes.jsonl; done
timeout 600 python -m pytest tests/test_gpu_timed_kernels.py -m gpu -q -k "bench or valu" > gpurun_out/r03j_pytest.log 2>&1; tail -5 gpurun_out/r03j_pytest.log
