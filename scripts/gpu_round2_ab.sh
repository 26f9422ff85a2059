cd $GRAFT_REPO_ROOT
echo "== current"; python scripts/window_sweep_time.py 2>&1 | tail -4
echo "== previous commit (K=4 build)"; (cd scripts/_build/prev && python scripts/window_sweep_time.py 2>&1 | tail -4)
echo "== current again"; python scripts/window_sweep_time.py 2>&1 | tail -4
for w in 4 8; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm window=$w', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['launch_ms'])"
done
timeout 300 python bench.py --model fm --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fm', d['ms_per_step'], d['value'])"
