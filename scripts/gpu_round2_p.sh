cd $GRAFT_REPO_ROOT
scripts/prof.sh r02_p_deepfm_kernel_stats --steps 800 --warmup 100 --no_cpu_baseline > /dev/null
head -12 gpurun_out/r02_p_deepfm_kernel_stats.txt | cut -c1-130
grep -A9 "timeline" gpurun_out/r02_p_deepfm_kernel_stats.txt | cut -c1-40,150-190
scripts/prof.sh r02_p_deepfm_plain_kernel_stats --no_overlap --steps 400 --warmup 50 --no_cpu_baseline > /dev/null
head -10 gpurun_out/r02_p_deepfm_plain_kernel_stats.txt | cut -c1-130
timeout 900 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_cin_bf16.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm bf16', d['ms_per_step'], d['value'])"
