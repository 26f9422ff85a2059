cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin_bf16.py tests/test_gpu_xdeepfm.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for r in 0 1; do
RSX_XDFM_SORT_RIDE=$r timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm bf16 ride=$r', d['ms_per_step'], d['value'])"
done
timeout 300 python bench.py --model xdeepfm --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm f32', d['ms_per_step'], d['value'])"
scripts/prof.sh r02_s_xdeepfm_bf16_kernel_stats --model xdeepfm --cin_bf16 --steps 400 --warmup 50 --no_cpu_baseline > /dev/null
head -24 gpurun_out/r02_s_xdeepfm_bf16_kernel_stats.txt | cut -c1-130
