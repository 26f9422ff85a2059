set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin_bf16.py -x -q -s 2>&1 | tail -25 | tee gpurun_out/r02_c_bf16_tests.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_end_to_end.py -x -q 2>&1 | tail -15 | tee gpurun_out/r02_c_fullsize.log
timeout 300 python bench.py --model xdeepfm --no_cpu_baseline 2>&1 | tail -1 | tee gpurun_out/r02_c_xdeepfm_f32.log
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | tee gpurun_out/r02_c_xdeepfm_bf16.log
scripts/prof.sh r02_c_xdeepfm_bf16_kernel_stats --model xdeepfm --cin_bf16 --steps 400 --warmup 50 --no_cpu_baseline
