#!/bin/bash
# usage (GPU box, via gpurun): scripts/round_end.sh <tag>     e.g. r04_z
# The end-of-round evidence set: per-config rocprofv3 kernel tables, the bench lines (default + the driver's --steps 20), the
# emulated-world table, the stand-alone kernel roofline table, and the large-batch scatter in parts (3 repeats).
tag=${1:-rXX_z}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root; mkdir -p gpurun_out
scripts/gpu.sh prof ${tag}_deepfm_kernel_stats --no_cpu_baseline --no_configs --steps 800 --warmup 96
scripts/gpu.sh prof ${tag}_fm_kernel_stats --model fm --no_cpu_baseline --no_configs --steps 800 --warmup 96
scripts/gpu.sh prof ${tag}_dcn_kernel_stats --model dcn --no_cpu_baseline --no_configs --steps 320 --warmup 32
scripts/gpu.sh prof ${tag}_xdeepfm_f32_kernel_stats --model xdeepfm --cin_split 0 --no_cpu_baseline --no_configs --steps 320 --warmup 32
scripts/gpu.sh prof ${tag}_xdeepfm_bf16_kernel_stats --model xdeepfm --cin_bf16 --no_cpu_baseline --no_configs --steps 320 --warmup 32
# the split-operand CIN modes: 4 = xdeepfm.py's default (two fp16 planes fwd / dX, three bf16 planes dW), 3 = three bf16 planes
scripts/gpu.sh prof ${tag}_xdeepfm_x4_kernel_stats --model xdeepfm --cin_split 4 --no_cpu_baseline --no_configs --steps 320 --warmup 32
scripts/gpu.sh prof ${tag}_xdeepfm_x3_kernel_stats --model xdeepfm --cin_split 3 --no_cpu_baseline --no_configs --steps 320 --warmup 32
scripts/gpu.sh pmc ${tag}_MFMA_BUSY_xdeepfm_x4 "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" --model xdeepfm --cin_split 4 --no_configs --no_cpu_baseline --steps 32 --warmup 16
python scripts/cin_split_probe.py 2>/dev/null | grep -v amdgpu > gpurun_out/${tag}_cin_split_probe.txt
scripts/gpu.sh prof ${tag}_din_kernel_stats --model din --no_cpu_baseline --no_configs --steps 160 --warmup 16
TAG=${tag}_default scripts/gpu.sh bench
TAG=${tag}_steps20 scripts/gpu.sh bench --steps 20
# per-rank compute of an N-rank step (peers = other resident batches), both sparse exchanges, all models
scripts/emulate_table.sh > /dev/null 2>&1; cp gpurun_out/emulate_table.txt gpurun_out/${tag}_emulate_world_all_models.txt
scripts/gpu.sh roofline > /dev/null; cp gpurun_out/kernel_roofline_table.txt gpurun_out/${tag}_kernel_roofline_table.txt
for r in 1 2 3; do echo "# repeat $r"; python scripts/scatter_large.py 4096 65536 2>/dev/null; done > gpurun_out/${tag}_scatter_large_3_repeats.txt
# the data-parallel code path through RCCL at world 1 in its default configuration (round 6: collectives through the C ABI,
# captured into the step's graphs), both exchanges, next to the single-replica step; then 50 xdeepfm.py soak runs
scripts/dp_world1_r06.sh > gpurun_out/${tag}_dp_world1_rccl.txt 2>&1
cat gpurun_out/${tag}_dp_world1_rccl.txt
# MFMA-busy of din.py's attention kernels (the forward runs on the bf16 matrix cores with split operands since round 6)
scripts/gpu.sh pmc ${tag}_MFMA_BUSY_din "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" --model din --no_configs --no_cpu_baseline --steps 32 --warmup 16
# the headline's HBM-side traffic, fresh (separate --pmc passes)
scripts/gpu.sh pmc ${tag}_FETCH_SIZE_deepfm "FETCH_SIZE" --no_configs --no_cpu_baseline --steps 64 --warmup 32
scripts/gpu.sh pmc ${tag}_WRITE_SIZE_deepfm "WRITE_SIZE" --no_configs --no_cpu_baseline --steps 64 --warmup 32
ls -la gpurun_out | grep $tag
