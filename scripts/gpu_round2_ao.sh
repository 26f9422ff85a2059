cd $GRAFT_REPO_ROOT
b() { timeout 300 python bench.py --model $2 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'])"; }
for i in 1 2; do
b base dcn
RSX_LIB_PATH=$PWD/scripts/_build/librsx_mw5.so b minw5 dcn
RSX_LIB_PATH=$PWD/scripts/_build/librsx_mw6.so b minw6 dcn
done
b base din
RSX_LIB_PATH=$PWD/scripts/_build/librsx_mw5.so b minw5 din
RSX_LIB_PATH=$PWD/scripts/_build/librsx_mw6.so b minw6 din
