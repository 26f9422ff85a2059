#!/bin/bash
# builds the -DRSX_STAMPS profiling variant of librsx.so for scripts/stamp_probe.py (never loaded by the product)
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/scripts/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -parallel-jobs=8 -ffp-contract=off -DRSX_STAMPS $STAMP_DEFS \
  -I$root/include -I$root/recsys_amd/csrc $root/recsys_amd/csrc/*.hip $root/recsys_amd/csrc/*.cpp -o $root/scripts/_build/${STAMP_OUT:-librsx_stamps.so}
