cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_cin_bf16.py tests/test_gpu_embedding.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm bf16', d['ms_per_step'], d['value'])"
scripts/prof.sh r02_r_xdeepfm_bf16_kernel_stats --model xdeepfm --cin_bf16 --steps 400 --warmup 50 --no_cpu_baseline > /dev/null
head -20 gpurun_out/r02_r_xdeepfm_bf16_kernel_stats.txt | cut -c1-130
