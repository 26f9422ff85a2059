cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py -x -q -k "dcn or cross or large_batch or dp" 2>&1 | tail -2
b() { python bench.py --no_cpu_baseline --no_configs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('launches_per_step'))"; }
for i in 1 2 3; do echo -n "dcn gather+cross fused: "; b --model dcn; echo -n "dcn two launches: "; RSX_GATHER_CROSS=0 b --model dcn; done
