cd $GRAFT_REPO_ROOT
python scripts/scatter_parts.py 256 2>&1 | tail -7; python scripts/scatter_parts.py 4096 2>&1 | tail -6
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5
for m in deepfm fm dcn xdeepfm din; do
timeout 300 python bench.py --model $m --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])"
done
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm bf16', d['ms_per_step'], d['value'])"
