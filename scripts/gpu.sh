#!/bin/bash
# The ONE GPU-box driver (run through gpurun from the repo root):  scripts/gpu.sh <task> [args] [-- <task> [args] ...]
# Tasks (outputs under gpurun_out/, copy what should be judged into profiles/):
#   test [pytest args]          pytest -m gpu (default: the whole GPU suite)
#   bench [bench args]          python bench.py ... -> gpurun_out/bench_<tag>.json   (tag = $TAG or "run")
#   prof <name> [bench args]    rocprofv3 --kernel-trace --stats of bench.py -> gpurun_out/<name>.txt
#   profcmd <name> <cmd...>     the same for an arbitrary python command
#   pmc <name> "<ctrs>" [bench args]      one --pmc pass of bench.py (kernel-trace only)
#   pmccmd <name> "<ctrs>" <cmd...>       the same for an arbitrary python command
#   py <script> [args]          python <script> ... | tee gpurun_out/<basename>.txt
#   emulate <model> [bench args]          bench.py --emulate_world 2 4 8 for one model -> gpurun_out/emulate_<model>.txt
#   roofline                    scripts/kernel_roofline.py -> gpurun_out/kernel_roofline_table.txt
# Several tasks are chained with a stand-alone "--".
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
run_task() {
  local task=$1; shift
  case $task in
    test)    if [ $# -eq 0 ]; then set -- tests -m gpu -x -q; fi
             timeout ${TEST_TIMEOUT:-2400} python -m pytest "$@" 2>&1 | tail -${TAIL:-15} ;;
    bench)   timeout 900 python bench.py "$@" 2> gpurun_out/bench_${TAG:-run}.err | tail -1 > gpurun_out/bench_${TAG:-run}.json
             python - gpurun_out/bench_${TAG:-run}.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("bench: no JSON line", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-3000:]); sys.exit(0)
print("bench", d["config"]["workload"][:60], "ms/step", d["ms_per_step"], "value", d["value"], "repeats", d["config"]["timed_repeats_ms_per_step"])
r = d.get("roofline") or {}
print("  roofline", {k: r.get(k) for k in ("kernel", "achieved", "frac", "launch_ms", "tf1_equivalent_step_frac", "moved_bytes_step_frac")})
for c in d.get("configs", []):
    print("  config", c.get("workload", "")[:50], c.get("ms_per_step"), c.get("examples_per_sec"), c.get("error"),
          {k: c.get(k) for k in ("tf1_equivalent_step_frac", "moved_bytes_step_frac", "cin_mfma_step_frac")})
if "cpu_baseline" in d:
    print("  cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
             ;;
    prof)    scripts/prof.sh "$@" ;;
    profcmd) scripts/prof_cmd.sh "$@" ;;
    pmc)     scripts/pmc.sh "$@" ;;
    pmccmd)  scripts/pmc_cmd.sh "$@" ;;
    py)      local s=$1; shift; timeout ${PY_TIMEOUT:-1200} python $s "$@" 2>&1 | tee gpurun_out/$(basename $s .py)${TAG:+_$TAG}.txt | tail -${TAIL:-40} ;;
    emulate) local m=$1; shift
             for n in 1 2 4 8; do
               if [ $n -eq 1 ]; then e=""; else e="--emulate_world $n"; fi
               timeout 600 python bench.py --model $m $e --no_cpu_baseline --no_configs "$@" 2>gpurun_out/emulate_$m.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
except Exception as ex:
    print('emulate $m world $n: no JSON line', ex); print(open('gpurun_out/emulate_$m.err').read()[-2500:]); sys.exit(0)
c = d['config']
print('emulated per-rank compute: model $m world $n exchange', c.get('dp_exchange', '-'), 'RSX_DP_EXCHANGE=${RSX_DP_EXCHANGE:-unique} peers', '$*' if '$*' else 'distinct batches', ' ms_per_step', d['ms_per_step'], ' adam_window', c['adam_window'], ' global_batch', $n * int(c['global_batch']), ' gradient bytes/rank', c.get('dp_gradient_bytes'), ' ids-phase bytes/rank', c.get('dp_ids_phase_bytes'), ' launches/step', c.get('launches_per_step'))"
             done | tee gpurun_out/emulate_$m${TAG:+_$TAG}.txt ;;
    roofline) timeout 1200 python scripts/kernel_roofline.py 2>&1 | tee gpurun_out/kernel_roofline_table.txt | tail -40 ;;
    *) echo "unknown task $task"; return 1 ;;
  esac
}
args=()
for x in "$@"; do
  if [ "$x" == "--" ]; then run_task "${args[@]}"; args=(); else args+=("$x"); fi
done
if [ ${#args[@]} -gt 0 ]; then run_task "${args[@]}"; fi
