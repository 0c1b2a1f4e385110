#!/bin/bash
# Same-box A/B builds (dev aid; nothing here ships).
#   tools/ab_build.sh lib <tag> [-DNAME=VALUE ...]   -> gym_pomdp_amd/_lib/libpomdp_hip_<tag>.so: the product library of the
#       working tree with extra defines (e.g. -DPOMDP_QUAD_MIN_LANES=4096); tools/gpu_small_shards.py and
#       tools/gpu_ab_bench.py take such variants by path
#   tools/ab_build.sh rev <revA> [revB=working tree] -> gym_pomdp_amd/_lib/libpomdp_hip_a.so / _b.so from two git revisions
#       (revisions from round 3 on: the library is built by gym_pomdp_amd/_native.py, one object per translation unit)
set -e
mode=$1; shift
build() {  # <source tree> <output .so> [defines...]
  local src=$1 out=$2; shift 2
  (cd $src && python - "$out" "$@" <<'PY'
import sys
sys.path.insert(0, ".")
from gym_pomdp_amd import _native
print(_native.build(force=True, out=sys.argv[1], defines=sys.argv[2:]))
PY
  )
}
if [ "$mode" = lib ]; then
  tag=$1; shift
  build $PWD $PWD/gym_pomdp_amd/_lib/libpomdp_hip_$tag.so "$@"
elif [ "$mode" = rev ]; then
  A=$1; B=${2:-WORK}
  one() {  # <rev> <tag>
    if [ "$1" = WORK ]; then SRC=$PWD; else SRC=/tmp/ab_$2; rm -rf $SRC; git worktree add -f $SRC $1 >/dev/null 2>&1; fi
    build $SRC $PWD/gym_pomdp_amd/_lib/libpomdp_hip_$2.so
    if [ "$1" != WORK ]; then git worktree remove --force $SRC; fi
  }
  one $A a
  one $B b
else
  echo "usage: tools/ab_build.sh lib <tag> [-D...] | rev <revA> [revB]"; exit 1
fi
ls -la gym_pomdp_amd/_lib/*.so
