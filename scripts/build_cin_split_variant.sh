#!/bin/bash
# usage: scripts/build_cin_split_variant.sh <name> [extra hipcc flags]   e.g.  stamps -DRSX_STAMPS   |   dbg1 -DRSX_CIN_DBG=1
# A probe build of ONE translation unit (csrc/cin_split.hip) linked with the product build's other objects (recsys_amd/_obj, run
# `python -m recsys_amd.build` first) -> scripts/_build/librsx_cs_<name>.so: seconds instead of the minutes of build_stamps.sh.
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $root/scripts/_build
# (the newest object of every other source: the object cache may still hold older builds)
objs=$(ls -t $root/recsys_amd/_obj/*.o | grep -v cin_split | awk -F/ '{n=$NF; sub(/\.[0-9a-f]+\.o$/, "", n); if (!(n in seen)) {seen[n]=1; print}}')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -I$root/include -I$root/recsys_amd/csrc \
  -c $root/recsys_amd/csrc/cin_split.hip -o $root/scripts/_build/cin_split_$name.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $root/scripts/_build/cin_split_$name.o -o $root/scripts/_build/librsx_cs_$name.so && \
rm -f $root/scripts/_build/cin_split_$name.o
