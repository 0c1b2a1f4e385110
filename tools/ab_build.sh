#!/bin/bash
# Same-box A/B builds (dev aid; nothing here ships).
#   tools/ab_build.sh lib <tag> [-DNAME=VALUE ...]   -> gym_pomdp_amd/_lib/libpomdp_hip_<tag>.so: the product library of the
#       working tree with extra defines (e.g. -DPOMDP_QUAD_MIN_LANES=4096); tools/gpu_small_shards.py and
#       tools/gpu_ab_bench.py take such variants by path
#   tools/ab_build.sh rev <revA> [revB=working tree] -> gym_pomdp_amd/_lib/libpomdp_hip_a.so / _b.so from two git revisions
set -e
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC"
mode=$1; shift
if [ "$mode" = lib ]; then
  tag=$1; shift
  $HIPCC "$@" -o gym_pomdp_amd/_lib/libpomdp_hip_$tag.so gym_pomdp_amd/csrc/pomdp_kernels.hip
  ls -la gym_pomdp_amd/_lib/libpomdp_hip_$tag.so
elif [ "$mode" = rev ]; then
  A=$1; B=${2:-WORK}
  build() {  # <rev> <tag>
    if [ "$1" = WORK ]; then SRC=$PWD; else SRC=/tmp/ab_$2; rm -rf $SRC; git worktree add -f $SRC $1 >/dev/null 2>&1; fi
    (cd $SRC && $HIPCC -o $OLDPWD/gym_pomdp_amd/_lib/libpomdp_hip_$2.so gym_pomdp_amd/csrc/pomdp_kernels.hip)
    if [ "$1" != WORK ]; then git worktree remove --force $SRC; fi
  }
  build $A a
  build $B b
  ls -la gym_pomdp_amd/_lib/libpomdp_hip_a.so gym_pomdp_amd/_lib/libpomdp_hip_b.so
else
  echo "usage: tools/ab_build.sh lib <tag> [-D...] | rev <revA> [revB]"; exit 1
fi
