#!/bin/bash
# A/B (GPU box): wave-tile configurations of cin_bwd_dw_bf16_k (RSX_CIN_DW16_CFG = 10 * FT + NT), xdeepfm.py --cin_bf16 step time
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${CFGS:-32 24 34 44 18 28 38}; do
  for rep in 1 2; do
    RSX_CIN_DW16_CFG=$cfg python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline --no_configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg $cfg xdeepfm bf16 ms/step', d['ms_per_step'], 'final_loss', d['config'].get('final_loss'))"
  done
done
