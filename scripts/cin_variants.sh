#!/bin/bash
# GPU box: A/B compile-time CIN variants on the stand-alone kernel timings
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $v -Iinclude -Irecsys_amd/csrc recsys_amd/csrc/*.hip recsys_amd/csrc/*.cpp -o recsys_amd/librsx.so 2>/dev/null
  echo "== $v"; python scripts/cin_probe.py 2>&1 | grep -v amdgpu.ids | grep bwd
done
