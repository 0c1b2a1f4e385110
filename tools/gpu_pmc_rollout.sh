export TMPDIR=/tmp
REPO=$PWD
rm -rf /tmp/pmcr; mkdir -p /tmp/pmcr; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/pmcr/p1 -o p1 -- python $REPO/bench.py --env rock15 --mode rollout --lanes-per-gpu 2097152 --steps 20 --warmup 10 > /tmp/pmcr/p1.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3,glob
db=glob.glob('/tmp/pmcr/p1/**/*_results.db',recursive=True)
c=sqlite3.connect(db[0])
rows=c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%rollout%' group by kernel_name, counter_name").fetchall()
for k,cn,n,a,d in rows: print(k[:50], cn, n, "%.1f"%a, "dur %.1f us"%(d/1e3))
PY
tail -1 /tmp/pmcr/p1.log | cut -c1-300
