set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_din.py tests/test_gpu_dp.py -x -q 2>&1 | tail -12 | tee gpurun_out/r02_l_tests.log
timeout 600 python scripts/kernel_roofline.py 2>&1 | grep -v amdgpu.ids | grep "segsum\|field_sort\|gather" | tee gpurun_out/r02_l_roofline.log
timeout 300 python bench.py --model dcn --no_cpu_baseline 2>&1 | tail -1 | cut -c1-250 | tee gpurun_out/r02_l_dcn.log
timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | cut -c1-250 | tee gpurun_out/r02_l_deepfm.log
