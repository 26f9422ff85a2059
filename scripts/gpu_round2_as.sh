cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_adam_window.py tests/test_gpu_end_to_end.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python scripts/e2e_train_bench.py 2>&1 | grep -v amdgpu | tail -10
RSX_LAUNCH_THREAD=1 timeout 300 python scripts/e2e_train_bench.py 2>&1 | grep -v amdgpu | grep "Estimator.train"
