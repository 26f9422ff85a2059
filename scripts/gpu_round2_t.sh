cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin_bf16.py tests/test_gpu_xdeepfm.py -x -q 2>&1 | tail -3
run() { RSX_XDFM_SWEEP_WEIGHTS=$1 timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w=$1', d['ms_per_step'], d['value'])"; }
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['value'])"
run 0,0,0,0,1,2,2,1,1,2,2
run 0,0,0,0,1,2,2,1,1,3,2
run 0,0,0,0,1,2,2,2,2,4,2
run 1,1,0,0,1,2,2,1,1,3,2
run 0,0,0,0,1,1.5,1.5,1.5,1.5,3,1.5
run 0,0,0,0,0.5,1,1,2,2,4,1
run 0,0,0,0,1,1,1,2,2,5,1
run 0,0,0,0,0,0,0,2,2,5,0
run 0,0,0,0,0,0,0,0,0,1,0
scripts/prof.sh r02_t_xdeepfm_bf16_kernel_stats --model xdeepfm --cin_bf16 --steps 400 --warmup 50 --no_cpu_baseline > /dev/null
head -16 gpurun_out/r02_t_xdeepfm_bf16_kernel_stats.txt | cut -c1-130
scripts/prof.sh r02_t_xdeepfm_bf16_plain_kernel_stats --model xdeepfm --cin_bf16 --no_overlap --steps 400 --warmup 50 --no_cpu_baseline > /dev/null
head -18 gpurun_out/r02_t_xdeepfm_bf16_plain_kernel_stats.txt | cut -c1-130
