cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_din.py -x -q 2>&1 | tail -2
b() { python bench.py --no_cpu_baseline --no_configs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do echo -n "din: "; b --model din; done
for i in 1 2 3; do echo -n "dcn new rule: "; b --model dcn; echo -n "dcn old rule: "; RSX_TOWER_BIG_MIN_K_BWD=256 b --model dcn; done
