set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_golden.py tests/test_gpu_din.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6 | tee gpurun_out/r02_i_tests.log
timeout 300 python scripts/stamp_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_i_stamps.log
timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | cut -c1-330 | tee gpurun_out/r02_i_deepfm.log
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/r02_i_xdeepfm_bf16.log
timeout 300 python bench.py --model xdeepfm --no_cpu_baseline 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/r02_i_xdeepfm_f32.log
for w in "1,2,0,0,1,1.5,1.5,2,1.5,1.5" "0.5,1,0,0,1,2,2,1.5,1,2" "0.5,1,0.5,0.5,1,1.5,1.5,1.5,1,1.5" "0,0,0,0,1,2,2,1,1,2"; do
  echo "xdfm weights $w"; RSX_XDFM_SWEEP_WEIGHTS=$w timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r02_i_xdfm_weights.log
done
