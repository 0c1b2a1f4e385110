#!/bin/bash
# Build tools/microbench from two git revisions for a same-box A/B run (dev aid).
# usage: tools/ab_build.sh <revA> [revB=working tree]  ->  tools/microbench_a, tools/microbench_b
set -e
A=$1; B=${2:-WORK}
build() {  # <rev> <out>
  if [ "$1" = WORK ]; then SRC=$PWD; else SRC=/tmp/ab_$2; rm -rf $SRC; git worktree add -f $SRC $1 >/dev/null 2>&1; fi
  (cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -o $OLDPWD/tools/microbench_$2 tools/microbench.hip)
  if [ "$1" != WORK ]; then git worktree remove --force $SRC; fi
}
build $A a
build $B b
ls -la tools/microbench_a tools/microbench_b
