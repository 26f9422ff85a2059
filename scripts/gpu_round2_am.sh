cd $GRAFT_REPO_ROOT
b() { timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
b new_dxg
RSX_TOWER_DXG=0 b new_nodxg
(cd scripts/_build/prev && b old_dxg)
(cd scripts/_build/prev && RSX_TOWER_DXG=0 b old_nodxg)
done
