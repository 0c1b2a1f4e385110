export TMPDIR=/tmp
REPO=/root/repo
rm -rf /tmp/pmch; mkdir -p /tmp/pmch; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/pmch/p1 -o p1 -- python $REPO/bench.py --mode heuristic --env $1 --prewarm 0 --warmup 128 --steps 512 > /tmp/pmch/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_BUSY_CYCLES -d /tmp/pmch/p2 -o p2 -- python $REPO/bench.py --mode heuristic --env $1 --prewarm 0 --warmup 128 --steps 512 > /tmp/pmch/p2.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3,glob
for sub in ("p1","p2"):
    db=glob.glob('/tmp/pmch/%s/**/*_results.db'%sub,recursive=True)
    if not db: print("no db", sub); continue
    c=sqlite3.connect(db[0])
    rows=c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%heuristic_steps%' group by kernel_name, counter_name").fetchall()
    for k,cn,n,a,d in rows: print(k[:60], cn, n, "%.1f"%a, "dur %.1f us"%(d/1e3))
PY
