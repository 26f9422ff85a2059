cd $GRAFT_REPO_ROOT
python scripts/window_sweep_time.py 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_adam_window.py -x -q 2>&1 | tail -15
