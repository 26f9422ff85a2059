cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_adam_window.py tests/test_gpu_dp.py tests/test_gpu_xdeepfm.py -x -q 2>&1 | tail -8
for m in fm dcn xdeepfm; do
for w in 1 8; do
RSX_ADAM_WINDOW=$w timeout 600 python bench.py --model $m --emulate_world 2 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m emulate_world 2 window=$w', d['ms_per_step'])"
done
done
