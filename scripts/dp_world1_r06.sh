#!/bin/bash
# GPU box (round 6): every model's data-parallel step through RCCL at world 1 in its DEFAULT configuration -- collectives issued
# through the C ABI on the step's own stream and captured into the step's graphs (dist.DirectComm) -- with the unique-list
# exchange forced on (the default from 2 ranks on), next to the single-replica step of the same build; then SOAK xdeepfm.py runs
# (round 5: torch's NCCL watchdog aborted captured runs one time in three) counting the ones that print their line.
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
port=29620
line() { python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    print('$1', ' ms_per_step', d['ms_per_step'], ' adam_window', c['adam_window'], ' launches/step', c.get('launches_per_step'), ' exchange', c.get('dp_exchange'), ' backend', d.get('backend'), ' workload', c['workload'][-90:])
except Exception as e:
    print('$1', 'NO LINE', e)"; }
for m in ${MODELS:-deepfm fm dcn xdeepfm din}; do
  timeout 600 python bench.py --model $m --no_cpu_baseline --no_configs "$@" 2>/dev/null | grep '"metric"' | tail -1 | line "single replica: model $m"
  for ex in unique examples; do
    port=$((port + 1))
    RSX_DP_EXCHANGE=$ex RSX_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus 1 --model $m --no_cpu_baseline --no_configs "$@" 2>/dev/null | grep '"metric"' | tail -1 | line "world-1 RCCL via C ABI, captured (default): model $m RSX_DP_EXCHANGE=$ex"
  done
done
ok=0; n=${SOAK:-50}
for i in $(seq 1 $n); do
  port=$((port + 1))
  out=$(RSX_DP_EXCHANGE=unique RSX_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus 1 --model xdeepfm --no_cpu_baseline --no_configs --steps 200 --warmup 48 --repeats 2 2>/dev/null | grep -c '"metric"')
  if [ "$out" = "1" ]; then ok=$((ok + 1)); fi
done
echo "soak: xdeepfm.py world-1 RCCL (C ABI, captured collectives, unique-list exchange): $ok of $n runs printed their line ($((n - ok)) aborted)"
