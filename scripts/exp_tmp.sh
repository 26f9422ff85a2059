cd $GRAFT_REPO_ROOT
b() { python bench.py --no_cpu_baseline --no_configs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('launches_per_step'), d['config']['timed_repeats_ms_per_step'])"; }
echo "dcn split=1"; b --model dcn
echo "dcn split=0"; RSX_SORT_SPLIT=0 b --model dcn
scripts/gpu.sh test 2>&1 | tail -5
