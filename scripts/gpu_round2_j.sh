set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_golden.py tests/test_gpu_din.py tests/test_gpu_fullsize.py tests/test_gpu_xdeepfm.py tests/test_gpu_dp.py -x -q 2>&1 | tail -6 | tee gpurun_out/r02_j_tests.log
timeout 300 python scripts/stamp_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_j_stamps.log
timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | cut -c1-330 | tee gpurun_out/r02_j_deepfm.log
RSX_TOWER_DXG=0 timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | cut -c1-200 | tee gpurun_out/r02_j_deepfm_nodxg.log
