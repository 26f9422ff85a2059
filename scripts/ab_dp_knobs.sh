#!/bin/bash
# A/B of the data-parallel step's knobs at an emulated world (distinct peer batches): scripts/ab_dp_knobs.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no_cpu_baseline --no_configs $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$tag', 'ms_per_step', d['ms_per_step'], 'window', c['adam_window'], 'launches', c.get('launches_per_step'))"; }
for m in ${MODELS:-deepfm dcn}; do
  ARGS="--model $m --emulate_world ${WORLD:-8}"
  run "$m N=${WORLD:-8} default            " A=1
  run "$m N=${WORLD:-8} NR=1 always        " RSX_WIN_NR4_MIN=1000000
  run "$m N=${WORLD:-8} window8            " RSX_FORMS=adam_window_large=8
  run "$m N=${WORLD:-8} window8 NR=1       " RSX_FORMS=adam_window_large=8 RSX_WIN_NR4_MIN=1000000
done
