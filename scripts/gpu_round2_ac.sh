cd $GRAFT_REPO_ROOT
echo "== current"; python scripts/window_sweep_time.py 2>&1 | tail -8
echo "== previous commit (K=4 build)"; (cd scripts/_build/prev && python scripts/window_sweep_time.py 2>&1 | tail -4)
timeout 1200 python -m pytest tests/test_gpu_adam_window.py tests/test_gpu_embedding.py -x -q 2>&1 | tail -5
for w in 1 4 8; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('deepfm window=$w', d['ms_per_step'], d['value'], r['achieved'], r['launch_ms'], r.get('single_step_sweep',{}).get('launch_ms'))"
done
timeout 300 python bench.py --model fm --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fm', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm bf16', d['ms_per_step'], d['value'])"
