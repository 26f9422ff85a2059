cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for big in 0 1; do
RSX_TOWER_BIG=$big timeout 300 python bench.py --model dcn --no_cpu_baseline --no_configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dcn RSX_TOWER_BIG=$big', d['ms_per_step'])"
done; done
for big in 0 1; do
RSX_TOWER_BIG=$big timeout 300 python bench.py --model din --no_cpu_baseline --no_configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('din RSX_TOWER_BIG=$big', d['ms_per_step'])"
done
