#!/bin/bash
# GPU box: per-rank compute of an N-rank step (bench.py --emulate_world, peers = other resident batches) for every model and both
# sparse exchanges -> gpurun_out/emulate_table.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
for ex in unique examples; do
  for m in ${MODELS:-deepfm fm dcn xdeepfm din}; do
    RSX_DP_EXCHANGE=$ex TAG=$ex scripts/gpu.sh emulate $m 2>&1 | grep -v "world 1 " 
  done
done | tee gpurun_out/emulate_table.txt
