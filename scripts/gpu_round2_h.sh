set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin_bf16.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r02_h_bf16_tests.log
timeout 300 python scripts/stamp_probe.py 2>&1 | grep -v amdgpu.ids | grep "head\|overlap" | tee gpurun_out/r02_h_stamps.log
scripts/prof.sh r02_h_xdeepfm_bf16_plain_kernel_stats --model xdeepfm --cin_bf16 --no_overlap --steps 200 --warmup 30 --no_cpu_baseline
scripts/prof.sh r02_h_xdeepfm_bf16_kernel_stats --model xdeepfm --cin_bf16 --steps 200 --warmup 30 --no_cpu_baseline
