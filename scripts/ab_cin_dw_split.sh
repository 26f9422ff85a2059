#!/bin/bash
# A/B: xdeepfm.py --cin_bf16 with the weight gradients as one workgroup per tile (0) or as a K-split GEMM tile (S slices)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in 0 4 2 8 0 4; do
  RSX_CIN_DW16_SPLIT=$s timeout 600 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline --no_configs --steps 256 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('xdeepfm --cin_bf16 RSX_CIN_DW16_SPLIT=$s ms_per_step', d['ms_per_step'], d['config']['timed_repeats_ms_per_step'])"
done
