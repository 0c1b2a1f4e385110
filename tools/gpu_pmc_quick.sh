#!/bin/bash
# One PMC pass over the fused launches of one env (dev aid): instructions per wave-step and the launch duration.
# usage: tools/gpu_pmc_quick.sh <env> [extra bench args]
ENV=${1:-network}; shift
export TMPDIR=/tmp
REPO=$PWD
rm -rf /tmp/pmcq; mkdir -p /tmp/pmcq
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/pmcq/p1 -o p1 -- python $REPO/bench.py --env $ENV --prewarm 0 --warmup 64 --steps 640 --seeds 0 --repeats 1 --no-cpu-baseline "$@" > /tmp/pmcq/p1.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3,glob
db=glob.glob('/tmp/pmcq/p1/**/*_results.db',recursive=True)
c=sqlite3.connect(db[0])
rows=c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%steps%kernel%' group by kernel_name, counter_name").fetchall()
ks={}
for k,cn,n,a,d in rows: ks.setdefault(k,{})[cn]=(n,a,d)
for k,v in ks.items():
    w=v['SQ_WAVES'][1]; d=v['SQ_WAVES'][2]
    print(k[:90])
    print('   launches %d  duration %.1f us = %.3f us/step (64 steps per launch)  waves %d'%(v['SQ_WAVES'][0], d/1e3, d/64e3, w))
    for cn in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS'):
        print('   %-20s %8.1f per wave-step'%(cn, v[cn][1]/w/64))
    wc=v['SQ_WAVE_CYCLES'][1]
    for cn in ('SQ_ACTIVE_INST_VALU','SQ_WAIT_INST_ANY','SQ_WAIT_ANY'):
        print('   %-20s %5.1f %% of wave cycles'%(cn, 100*v[cn][1]/wc))
PY
