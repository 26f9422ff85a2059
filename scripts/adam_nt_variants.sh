#!/bin/bash
# GPU box: sweep with / without non-temporal loads+stores, end-to-end on the models whose MFMA launches carry the sweep
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for v in "-DRSX_ADAM_NT=1" "-DRSX_ADAM_NT=0"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $v -Iinclude -Irecsys_amd/csrc recsys_amd/csrc/*.hip recsys_amd/csrc/*.cpp -o recsys_amd/librsx.so 2>/dev/null
  echo "== $v"
  for m in xdeepfm deepfm; do python bench.py --model $m --no_cpu_baseline 2>&1 | tail -1 | cut -c1-140; done
  RSX_CIN_DW_CFG=2 python bench.py --model xdeepfm --no_cpu_baseline 2>&1 | tail -1 | cut -c1-140
done
