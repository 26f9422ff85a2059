cd $GRAFT_REPO_ROOT
python scripts/window_sweep_time.py 2>&1 | tail -4
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
for w in 4 6 8; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm window=$w', d['ms_per_step'], d['value'])"
done
for m in fm xdeepfm; do
timeout 300 python bench.py --model $m --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])"
done
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm bf16', d['ms_per_step'], d['value'])"
for w in 4 8; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --model dcn --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dcn window=$w', d['ms_per_step'], d['value'])"
done
