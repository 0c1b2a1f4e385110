#!/bin/bash
# round 4, second GPU call: the whole GPU suite, the bench lines (default, the driver's form, per layout), the headline's
# kernel trace under rocprofv3 and the headline PMC set
set -x
T=${1:-r04b}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/$T/pytest_gpu.log
cat gpurun_out/$T/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/$T/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/$T/bench_steps20.err | tail -1 > gpurun_out/$T/bench_steps20.json
timeout 600 python bench.py 2>gpurun_out/$T/bench.err | tail -1 > gpurun_out/$T/bench.json
python - <<PY
import json
for f in ("bench_steps20", "bench"):
    try:
        d = json.load(open("gpurun_out/$T/%s.json" % f))
        r = d["roofline"]
        print(f, "value %.3e ms/step %.4g kernel_ms %.4g bound %s frac %.3f hbm %.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["bound"], r["frac"], r["hbm"]["frac"]))
        for k, v in (d.get("layouts") or {}).items():
            print("   layout", k, "%.3e kernel_ms %.4g hbm %.3f" % (v["value"], v["kernel_ms"], v["roofline"]["hbm_frac"]))
        for k, v in (d.get("configs") or {}).items():
            print("   config", k, "%.3e kernel_ms %.4g %s %s" % (v["value"], v["kernel_ms"], v["roofline"]["bound"], v["roofline"]["frac"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
cd /tmp
rm -rf /tmp/tr_$T; mkdir -p /tmp/tr_$T
rocprofv3 --kernel-trace --stats -d /tmp/tr_$T/trace -o t -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/tr_$T/bench_traced.log 2>&1
cd $OLDPWD
(echo "# python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline under rocprofv3 --kernel-trace --stats (MI355X)"; python tools/rocpd_summary.py /tmp/tr_$T | awk '{ if (substr($0,1,1)=="{") print; else print substr($0,1,300) }'; tail -1 /tmp/tr_$T/bench_traced.log) > gpurun_out/$T/headline_steps20.txt
head -30 gpurun_out/$T/headline_steps20.txt
bash tools/gpu_pmc_valu.sh ${T}_valu headline > gpurun_out/$T/pmc_valu.log 2>&1
tail -30 gpurun_out/$T/pmc_valu.log
