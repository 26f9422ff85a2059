#!/bin/bash
# A/B (single replica): the lazy window pass of segsum_adam_k as a dense grid (0) or a compact grid-stride walk (N = workgroups)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in deepfm fm dcn; do
for c in 0 256 512 0 256 512; do
  RSX_WIN_COMPACT=$c timeout 600 python bench.py --model $m --no_cpu_baseline --no_configs --steps 512 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$m RSX_WIN_COMPACT=$c ms_per_step', d['ms_per_step'], d['config']['timed_repeats_ms_per_step'])"
done; done
