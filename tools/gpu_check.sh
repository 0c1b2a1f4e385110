#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprof kernel trace.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 300 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
