#!/bin/bash
# Where does cin_split_fwd_k spend a step?  Probe builds of librsx.so with one ingredient of the field loop removed (RSX_CIN_DBG, see
# csrc/cin_split.hip; the results of those builds are WRONG, only their timing means something):
#   build (here, no GPU):  scripts/cin_split_where.sh build     -> scripts/_build/librsx_dbg{0,1,2,3}.so
#   run (GPU box):         scripts/cin_split_where.sh           -> stand-alone forward timings per build
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "build" ]; then
  for d in 0 1 2 3; do
    STAMP_OUT=librsx_dbg$d.so STAMP_DEFS="-URSX_STAMPS -DRSX_CIN_DBG=$d" bash $root/scripts/build_stamps.sh &
  done
  wait
  exit 0
fi
cd $root
for d in 0 1 2 3; do
  for e in 8 4; do
    echo "== RSX_CIN_DBG=$d (1: no MFMA, 2: no global filter loads in the loop, 3: no barrier in the loop)  E=$e"
    RSX_CIN_SPLIT_E=$e RSX_LIB_PATH=$root/scripts/_build/librsx_dbg$d.so python scripts/cin_split_probe.py 2>&1 | grep "split ns=3"
  done
done
