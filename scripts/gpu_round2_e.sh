set -x
cd $GRAFT_REPO_ROOT
timeout 600 python tests/tools/dbg_dcn4096.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_e_dbg_dcn.log
scripts/prof.sh r02_e_xdeepfm_bf16_plain_kernel_stats --model xdeepfm --cin_bf16 --no_overlap --steps 200 --warmup 30 --no_cpu_baseline
scripts/prof.sh r02_e_xdeepfm_f32_plain_kernel_stats --model xdeepfm --no_overlap --steps 200 --warmup 30 --no_cpu_baseline
bash scripts/gpu_round2_d.sh
