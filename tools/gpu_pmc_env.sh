#!/bin/bash
# PMC counters for one env's step kernels (dev aid). usage: tools/gpu_pmc_env.sh <env> [extra bench args]
ENV=${1:-battleship}; shift
export TMPDIR=/tmp
REPO=$PWD
mkdir -p /tmp/pmc
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/pmc/p1 -o p1 -- python $REPO/bench.py --env $ENV --prewarm 0 --warmup 64 --steps 640 --no-cpu-baseline "$@" > /tmp/pmc/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pmc/p2 -o p2 -- python $REPO/bench.py --env $ENV --prewarm 0 --warmup 64 --steps 640 --no-cpu-baseline "$@" > /tmp/pmc/p2.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3,glob
for sub in ('p1','p2'):
    db=glob.glob(f'/tmp/pmc/{sub}/**/*_results.db',recursive=True)
    if not db: print('no db',sub); continue
    c=sqlite3.connect(db[0])
    rows=c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%step%_kernel%' group by kernel_name, counter_name").fetchall()
    ks={}
    for k,cn,n,a,d in rows: ks.setdefault(k,{})[cn]=(n,a,d)
    for k,v in ks.items():
        print(k[:110])
        for cn,(n,a,d) in sorted(v.items()): print('    %-22s n=%d avg=%.1f dur_ns=%.0f'%(cn,n,a,d))
PY
