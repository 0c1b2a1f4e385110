#!/bin/bash
# dev aid: VALU / wait counters of the heuristic loop for library variants (tools/ab_build.sh): tools/heur_pmc.sh <env> <tag...>
export TMPDIR=/tmp
REPO=$PWD
env=$1; shift
cd /tmp
for lib in "$@"; do
  d=/tmp/hp_${lib}_$env; rm -rf $d
  GYM_POMDP_AMD_LIB=$REPO/gym_pomdp_amd/_lib/libpomdp_hip_$lib.so timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $d -o p -- python $REPO/bench.py --env $env --mode heuristic --steps 1024 > $d.log 2>&1
  python - $d $lib $env <<'PY'
import sys, glob, os, sqlite3
d, lib, env = sys.argv[1:4]
db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
c = sqlite3.connect(db)
v = {}
for k, cn in c.execute("select distinct kernel_name, counter_name from counters_collection where kernel_name like '%heuristic_steps%'"):
    rows = c.execute("select value, duration from counters_collection where kernel_name=? and counter_name=? order by start desc limit 4", (k, cn)).fetchall()
    v[cn] = sum(x[0] for x in rows) / len(rows); v["dur_us"] = sum(x[1] for x in rows) / len(rows) / 1e3
print(lib, env, " ".join("%s=%.4g" % kv for kv in sorted(v.items())))
print("   VALU per wave-step(256) %.1f  SALU %.1f  LDS %.1f  wave_cycles/busy_cycles %.2f  active_valu/busy %.3f" % (
    v["SQ_INSTS_VALU"] / v["SQ_WAVES"] / 256, v["SQ_INSTS_SALU"] / v["SQ_WAVES"] / 256,
    v["SQ_INSTS_LDS"] / v["SQ_WAVES"] / 256, v["SQ_WAVE_CYCLES"] / max(v["SQ_BUSY_CYCLES"], 1), v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_BUSY_CYCLES"], 1)))
PY
done
