set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin_bf16.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r02_g_bf16_tests.log
scripts/prof.sh r02_g_xdeepfm_bf16_plain_kernel_stats --model xdeepfm --cin_bf16 --no_overlap --steps 200 --warmup 30 --no_cpu_baseline
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/r02_g_xdeepfm_bf16.log
timeout 300 python scripts/stamp_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_g_stamps.log
