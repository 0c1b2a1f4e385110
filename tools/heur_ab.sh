# dev aid: bench.py --mode heuristic per library variant (tools/ab_build.sh): tools/heur_ab.sh "<envs>" <tag...>
envs=$1; shift
for lib in "$@"; do
  for env in $envs; do
    echo -n "$lib $env: "
    GYM_POMDP_AMD_LIB=$PWD/gym_pomdp_amd/_lib/libpomdp_hip_$lib.so python bench.py --env $env --mode heuristic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us/step  %.3e' % (d['ms_per_step']*1e3, d['value']))"
  done
done
