cd $GRAFT_REPO_ROOT
python scripts/window_sweep_time.py 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_adam_window.py tests/test_gpu_embedding.py tests/test_gpu_adam.py -x -q 2>&1 | tail -5
for w in 1 4; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm window=$w', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done
for m in fm dcn xdeepfm din; do
timeout 300 python bench.py --model $m --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])"
done
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm bf16', d['ms_per_step'], d['value'])"
