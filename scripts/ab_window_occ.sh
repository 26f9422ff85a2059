#!/bin/bash
# A/B (GPU box): adam_window_k compiled for 4 waves per SIMD (128 registers, spills at NW = 7) vs 3 (168 registers)
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== default build (RSX_ADAM_WIN_OCC=4)"; python scripts/window_sweep_time.py 2>&1 | tail -6
echo "== RSX_ADAM_WIN_OCC=3 build"; RSX_LIB_PATH=scripts/_build/librsx_winocc3.so python scripts/window_sweep_time.py 2>&1 | tail -6
for rep in 1 2; do
  python bench.py --no_cpu_baseline --no_configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm occ4', d['ms_per_step'])"
  RSX_LIB_PATH=scripts/_build/librsx_winocc3.so python bench.py --no_cpu_baseline --no_configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm occ3', d['ms_per_step'])"
done
