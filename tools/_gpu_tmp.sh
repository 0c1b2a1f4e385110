set -x
mkdir -p gpurun_out/r04f; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_timed_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "battleship" 2>&1 | tail -4 > gpurun_out/r04f/pytest_bs.log; cat gpurun_out/r04f/pytest_bs.log
for e in battleship5 battleship; do for l in packed columns; do timeout 300 python bench.py --env $e --layout $l --no-cpu-baseline --no-extras --seeds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e $l', 'value %.3e kernel %.4g us/step, hbm %.3f' % (d['value'], r['kernel_ms']*1e3, r['hbm']['frac']))
"; done; done 2>&1 | grep value | tee gpurun_out/r04f/bench_bs.txt
timeout 300 python bench.py --env battleship --lanes-per-gpu 524288 --no-cpu-baseline --no-extras --seeds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('battleship 2^19 packed', 'value %.3e kernel %.4g us/step' % (d['value'], r['kernel_ms']*1e3))
" | tee -a gpurun_out/r04f/bench_bs.txt
