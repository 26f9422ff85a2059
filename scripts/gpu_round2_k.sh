set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 | tee gpurun_out/r02_k_tests.log
timeout 300 python scripts/stamp_probe.py 2>&1 | grep -v amdgpu.ids | grep "bwd\|overlap" | tee gpurun_out/r02_k_stamps.log
for w in "0,0,2,3,3,2" "0,0,1.5,3,3,2.5" "0,0,1,3,3,3" "0,0,1.5,3.5,3.5,2" "0,0,1,3.5,3.5,2.5" "0,0,2,2.5,3.5,2"; do
  echo "weights $w"; RSX_SWEEP_WEIGHTS=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r02_k_weights.log
done
