set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_golden.py -x -q 2>&1 | tail -5
timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | cut -c1-330 | tee gpurun_out/r02_d_deepfm_default.log
for w in "0,1,2,3,3,2" "0,1.5,2,3,3,2" "0,1,2,3,3,3" "0,0,1.5,3,3,2.5" "0,0,2,3.5,3.5,2"; do
  echo "weights $w"; RSX_SWEEP_WEIGHTS=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r02_d_weights.log
done
scripts/prof.sh r02_d_deepfm_plain_path_kernel_stats --no_overlap --steps 400 --warmup 50 --no_cpu_baseline
scripts/prof.sh r02_d_deepfm_kernel_stats --steps 800 --warmup 100 --no_cpu_baseline
