#!/bin/bash
# GPU box: A/B the stand-alone Adam sweep across compile-time variants (timing only)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for v in "-DRSX_ADAM_U=4 -DRSX_ADAM_NT=0" "-DRSX_ADAM_U=8 -DRSX_ADAM_NT=0" "-DRSX_ADAM_U=2 -DRSX_ADAM_NT=0" "-DRSX_ADAM_U=4 -DRSX_ADAM_NT=1" "-DRSX_ADAM_U=8 -DRSX_ADAM_NT=1" "-DRSX_ADAM_U=1 -DRSX_ADAM_NT=0"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $v -Iinclude -Irecsys_amd/csrc recsys_amd/csrc/*.hip recsys_amd/csrc/*.cpp -o recsys_amd/librsx.so 2>/dev/null
  echo "== $v"; python scripts/adam_sweep.py 2>&1 | grep "rows=  840646\|rows= 3200000"
done
