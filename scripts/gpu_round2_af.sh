cd $GRAFT_REPO_ROOT
for side in 0 1; do
RSX_WIN_SIDE=$side timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm side=$side', d['ms_per_step'], d['value'])"
done
RSX_WIN_SIDE=1 RSX_WIN_SIDE_PRIO=1 timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm side=1 prio=1', d['ms_per_step'], d['value'])"
RSX_WIN_SIDE=1 RSX_WIN_SIDE_PRIO=-1 timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm side=1 prio=-1', d['ms_per_step'], d['value'])"
RSX_WIN_SIDE=1 timeout 600 python -m pytest tests/test_gpu_adam_window.py -x -q -k "resident" 2>&1 | tail -3
RSX_WIN_SIDE=1 scripts/prof.sh r02_af_deepfm_side --steps 400 --warmup 48 --no_cpu_baseline > /dev/null
grep -n "timeline sample" -A24 gpurun_out/r02_af_deepfm_side.txt | cut -c1-60,150-200
