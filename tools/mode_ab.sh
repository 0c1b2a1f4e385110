# dev aid: bench.py --mode <mode> per library variant (tools/ab_build.sh): tools/mode_ab.sh <mode> "<envs>" <tag...>
mode=$1; envs=$2; shift 2
for lib in "$@"; do
  for env in $envs; do
    echo -n "$lib $mode $env: "
    GYM_POMDP_AMD_LIB=$PWD/gym_pomdp_amd/_lib/libpomdp_hip_$lib.so python bench.py --env $env --mode $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  %.4e' % (d['ms_per_step'], d['value']))"
  done
done
