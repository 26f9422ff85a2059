cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_din.py tests/test_gpu_fullsize.py -x -q -k "din" 2>&1 | tail -2
b() { python bench.py --no_cpu_baseline --no_configs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do echo -n "din skip: "; b --model din; echo -n "din no skip: "; RSX_DIN_GATHER_SKIP=0 b --model din; done
