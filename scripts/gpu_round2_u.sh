cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_adam_window.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_embedding.py -x -q 2>&1 | tail -3
for w in 1 2 4; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm window=$w', d['ms_per_step'], d['value'])"
done
