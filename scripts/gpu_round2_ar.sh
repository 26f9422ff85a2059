cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q -m gpu -k "cross or dcn" 2>&1 | tail -4
b() { timeout 300 python bench.py --model $2 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'])"; }
for i in 1 2; do
RSX_CROSS_BWD4=0 b onewave dcn
b bwd4 dcn
done
RSX_CROSS_EPW4=1 b bwd4_epw1 dcn
RSX_CROSS_EPW4=4 b bwd4_epw4 dcn
timeout 600 python scripts/kernel_roofline.py 2>&1 | grep cross
