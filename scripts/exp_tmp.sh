cd $GRAFT_REPO_ROOT
b() { python bench.py --no_cpu_baseline --no_configs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('launches_per_step'))"; }
for i in 1 2; do for j in end mlp poolbwd; do echo -n "din join at $j: "; RSX_DIN_JOIN_AT=$j b --model din; done; done
