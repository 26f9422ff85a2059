cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_adam_window.py -x -q -m gpu -k "streaming" 2>&1 | tail -3
timeout 200 python scripts/e2e_train_bench.py 2>&1 | grep -v amdgpu | tail -8
RSX_LAUNCH_THREAD=0 timeout 200 python scripts/e2e_train_bench.py 2>&1 | grep -v amdgpu | grep "deepfm\|fm  "
timeout 400 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_dp_run_main.py -x -q -m gpu 2>&1 | tail -3
