#!/bin/bash
# GPU box: every model's data-parallel step through RCCL at world 1 (segmented graphs + eager collectives), both exchanges
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
port=29520
for ex in unique examples; do
for m in ${MODELS:-deepfm fm dcn xdeepfm din}; do
  port=$((port + 1))
  RSX_DP_EXCHANGE=$ex RSX_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus 1 --model $m --no_cpu_baseline --no_configs "$@" 2>/dev/null | grep '"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('world-1 RCCL: model $m exchange', c.get('dp_exchange'), ' ms_per_step', d['ms_per_step'], ' adam_window', c['adam_window'], ' launches/step', c.get('launches_per_step'), ' gradient bytes/rank', c.get('dp_gradient_bytes'), ' ids-phase bytes/rank', c.get('dp_ids_phase_bytes'))"
done; done
