#!/bin/bash
# Which hardware counters tell a fast trajectory buffer from a slow one?  (dev aid)  Runs tools/gpu_placement_probe.py (8 fresh
# allocations, 60 launches of 64 steps each) under rocprofv3 --kernel-trace --pmc, one pass per counter group, and prints per
# buffer the mean launch time and counter values.   usage: tools/gpu_placement_pmc.sh  -> gpurun_out/placement_pmc.txt
export TMPDIR=/tmp
REPO=$PWD
W=/tmp/ppmc; rm -rf $W; mkdir -p $W $REPO/gpurun_out
cd /tmp
G1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
G2="TCC_EA0_WRREQ_STALL_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"
G3="TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  PP_ALLOC_ONLY=1 PP_ALLOCS=8 PP_ALLOC_STAGGERS=4096 PP_ROUNDS=1 timeout 600 rocprofv3 --kernel-trace --pmc $G -d $W/p$i -o p -- python $REPO/tools/gpu_placement_probe.py 64 > $W/p$i.log 2>&1
  grep "stagger  " $W/p$i.log
done
python - <<'PY' > $REPO/gpurun_out/placement_pmc.txt
import glob, sqlite3, collections
for i in (1, 2, 3):
    f = glob.glob("/tmp/ppmc/p%d/**/*_results.db" % i, recursive=True)
    if not f:
        print("pass", i, "no db"); continue
    c = sqlite3.connect(f[0])
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    key = "dispatch_id" if "dispatch_id" in cols else cols[0]
    rows = c.execute("select %s, kernel_name, counter_name, value from counters_collection where kernel_name like '%%steps_quad_kernel%%' order by %s" % (key, key)).fetchall()
    kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    per = collections.OrderedDict()
    for d, k, cn, v in rows:
        per.setdefault(d, {})[cn] = per.setdefault(d, {}).get(cn, 0) + v
    ids = list(per)
    dur = {}
    if "dispatch_id" in kcols:
        for d, t in c.execute("select dispatch_id, duration from kernels where name like '%steps_quad_kernel%'"):
            dur[d] = t / 1e3
    print("# pass %d: %d dispatches of steps_quad_kernel, counters %s (columns of counters_collection: %s)" % (i, len(ids), sorted({cn for v in per.values() for cn in v}), cols))
    # 8 buffers x 60 launches (+ the launch that binds the buffer): split the dispatch sequence in 8 equal groups
    n = len(ids) // 8
    for b in range(8):
        g = ids[b * n:(b + 1) * n][n // 4:]          # drop the first quarter of each group (warm-up)
        names = sorted({cn for d in g for cn in per[d]})
        line = "  buffer %d: " % b
        if dur:
            line += "%.1f us  " % (sum(dur.get(d, 0) for d in g) / len(g))
        line += "  ".join("%s %.4g" % (cn, sum(per[d].get(cn, 0) for d in g) / len(g)) for cn in names)
        print(line)
PY
cat $REPO/gpurun_out/placement_pmc.txt
