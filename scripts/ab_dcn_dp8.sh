cd /root/repo
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no_cpu_baseline --no_configs --model dcn --emulate_world 8 --steps 256 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
run "default(merge256)     " A=1

run "NR=1                  " RSX_WIN_NR4_MIN=1000000



run "parts 8               " RSX_UX_PARTS=8
run "parts 2               " RSX_UX_PARTS=2
