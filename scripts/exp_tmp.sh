cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_mlp_fused.py -x -q 2>&1 | tail -3
python scripts/stamp_probe_mlp.py 2>&1 | tail -14
b() { python bench.py --no_cpu_baseline --no_configs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('launches_per_step'), d['config']['timed_repeats_ms_per_step'])"; }
echo "din fused mlp"; b --model din
