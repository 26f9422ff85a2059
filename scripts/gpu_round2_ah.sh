cd $GRAFT_REPO_ROOT
scripts/pmc.sh r02_ah_MFMA_BUSY_xdeepfm_bf16 SQ_VALU_MFMA_BUSY_CYCLES --model xdeepfm --cin_bf16 --no_cpu_baseline --steps 64 --warmup 16 > /dev/null
cat gpurun_out/pmc_r02_ah_MFMA_BUSY_xdeepfm_bf16.txt | cut -c1-150
RSX_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm dp world1 rccl', d['ms_per_step'], d['value'])"
for n in 2 8; do
timeout 600 python bench.py --emulate_world $n --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm emulate_world $n', d['ms_per_step'])"
done
