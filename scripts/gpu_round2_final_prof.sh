# GPU box: the end-of-round-2 evidence set (kernel tables of the default bench command per BASELINE config, PMC traffic of
# the sweep kernels, stand-alone roofline table, ingest rates).  Everything lands in gpurun_out/r02_z_*.
cd $GRAFT_REPO_ROOT
scripts/prof.sh r02_z_deepfm_kernel_stats --steps 800 --warmup 96 --no_cpu_baseline > /dev/null
scripts/prof.sh r02_z_fm_kernel_stats --model fm --steps 800 --warmup 96 --no_cpu_baseline > /dev/null
scripts/prof.sh r02_z_dcn_kernel_stats --model dcn --steps 400 --warmup 48 --no_cpu_baseline > /dev/null
scripts/prof.sh r02_z_xdeepfm_f32_kernel_stats --model xdeepfm --steps 400 --warmup 48 --no_cpu_baseline > /dev/null
scripts/prof.sh r02_z_xdeepfm_bf16_kernel_stats --model xdeepfm --cin_bf16 --steps 400 --warmup 48 --no_cpu_baseline > /dev/null
scripts/prof.sh r02_z_din_kernel_stats --model din --steps 200 --warmup 32 --no_cpu_baseline > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  scripts/pmc.sh r02_z_${c}_deepfm $c --no_cpu_baseline --steps 64 --warmup 16 > /dev/null
done
timeout 900 python scripts/kernel_roofline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_z_kernel_roofline_table.txt
timeout 600 python scripts/ingest_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_z_ingest_bench.txt
python scripts/window_sweep_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_z_window_sweep_time.txt
timeout 600 python scripts/e2e_train_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_z_e2e_train_bench.txt
timeout 300 python scripts/e2e_host_timeline.py 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r02_z_e2e_train_bench.txt
timeout 300 python scripts/graph_launch_cost.py 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r02_z_e2e_train_bench.txt
timeout 300 python scripts/ingest_layers.py 600000 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_z_ingest_layers.txt
timeout 600 python scripts/eval_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_z_eval_bench.txt
timeout 300 python bench.py --host_input --steps 800 --warmup 96 --no_cpu_baseline 2>&1 | tail -1 > gpurun_out/r02_z_bench_host_input.json
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r02_z_bench_default.json
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r02_z_bench_steps20.json
for f in deepfm fm dcn xdeepfm_f32 xdeepfm_bf16 din; do head -14 gpurun_out/r02_z_${f}_kernel_stats.txt | cut -c1-120; done
cat gpurun_out/pmc_r02_z_FETCH_SIZE_deepfm.txt gpurun_out/pmc_r02_z_WRITE_SIZE_deepfm.txt | cut -c1-150
cat gpurun_out/r02_z_window_sweep_time.txt; tail -12 gpurun_out/r02_z_ingest_bench.txt | cut -c1-150; cat gpurun_out/r02_z_e2e_train_bench.txt gpurun_out/r02_z_ingest_layers.txt gpurun_out/r02_z_eval_bench.txt | cut -c1-220
