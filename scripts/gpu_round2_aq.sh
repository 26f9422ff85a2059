cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_adam_window.py tests/test_gpu_fast_math.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python scripts/window_sweep_time.py 2>&1 | tail -6
b() { timeout 300 python bench.py --model $2 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'], d['roofline'].get('launch_ms'), d['roofline'].get('frac'))"; }
b fast deepfm
b fast deepfm
b fast fm
