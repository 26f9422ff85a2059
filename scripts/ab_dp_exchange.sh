cd /root/repo
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no_cpu_baseline --no_configs $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$tag', 'ms_per_step', d['ms_per_step'], 'window', c['adam_window'], 'launches', c.get('launches_per_step'))"; }
for m in deepfm fm dcn; do
  ARGS="--model $m --emulate_world 8"
  run "$m N=8 default            " A=1
  run "$m N=8 capture            " RSX_DP_CAPTURE=1
  run "$m N=8 window8            " RSX_FORMS=adam_window_large=8
  run "$m N=8 window8+capture    " RSX_FORMS=adam_window_large=8 RSX_DP_CAPTURE=1
  run "$m N=8 examples (legacy)  " RSX_DP_EXCHANGE=examples
  ARGS="--model $m --emulate_world 8 --emulate_identical"
  run "$m N=8 examples identical " RSX_DP_EXCHANGE=examples
  run "$m N=8 unique identical   " A=1
done
