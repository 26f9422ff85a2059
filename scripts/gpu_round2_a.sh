set -x
cd $GRAFT_REPO_ROOT
nproc; python -c "import os;print(os.cpu_count())"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | tail -2 | tee gpurun_out/r02_a_bench_driver_cmd.log
timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -2 | tee gpurun_out/r02_a_bench_default.log
for B in 4096 65536; do
  scripts/pmc_cmd.sh r02_a_scatter_B${B}_FETCH "FETCH_SIZE" scripts/segsum_prof.py $B
  scripts/pmc_cmd.sh r02_a_scatter_B${B}_WRITE "WRITE_SIZE" scripts/segsum_prof.py $B
  scripts/pmc_cmd.sh r02_a_scatter_B${B}_SQ "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" scripts/segsum_prof.py $B
done
