set -x
cd $GRAFT_REPO_ROOT
scripts/prof_cmd.sh r02_m_segsum_B4096 scripts/segsum_prof.py 4096
scripts/prof_cmd.sh r02_m_segsum_B65536 scripts/segsum_prof.py 65536
timeout 900 python -m pytest tests/test_gpu_cin_bf16.py tests/test_gpu_xdeepfm.py -x -q 2>&1 | tail -6 | tee gpurun_out/r02_m_tests.log
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-250 | tee gpurun_out/r02_m_xdeepfm_bf16.log
for w in "0,0,0,0,1,2,2,2,2" "0,0,0,0,1,2,2,3,2" "0,0,0,0,1,1.5,1.5,3,2.5" "0,0,0,0,0.5,2,2,3,2" "0,0,0,0,1,2,2,4,1.5"; do
  echo "xdfm weights $w"; RSX_XDFM_SWEEP_WEIGHTS=$w timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r02_m_xdfm_weights.log
done
scripts/prof.sh r02_m_xdeepfm_bf16_kernel_stats --model xdeepfm --cin_bf16 --steps 400 --warmup 50 --no_cpu_baseline
