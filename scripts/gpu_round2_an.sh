cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== pipelined HB=2"; python scripts/window_sweep_time.py 2>&1 | grep "window [248]"
echo "== pipelined HB=1"; RSX_LIB_PATH=$PWD/scripts/_build/librsx_hb1.so python scripts/window_sweep_time.py 2>&1 | grep "window [248]"
echo "== previous (phased, HB=4)"; RSX_LIB_PATH=$PWD/scripts/_build/librsx_prev.so python scripts/window_sweep_time.py 2>&1 | grep "window [248]"
done
