#!/bin/bash
# GPU box: build librsx.so variants with -DRSX_ABLATE=k and time the DeepFM step (results are INVALID: timing only)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DRSX_ABLATE=$k -Iinclude -Irecsys_amd/csrc recsys_amd/csrc/*.hip recsys_amd/csrc/*.cpp -o recsys_amd/librsx.so 2>/dev/null
  echo "ABLATE=$k"; scripts/prof.sh abl$k --no_cpu_baseline --steps 400 --warmup 40 | cut -c1-120; grep "tower_bwd_k\|tower_head_k\|segsum" gpurun_out/abl$k.txt | head -3
done
