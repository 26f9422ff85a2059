cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py -x -q 2>&1 | tail -4
for w in "0,0,1,3.5,3.5,2.5" "1,1,1,3,3,2.5" "0.5,0.5,1,3,3,2.5" "1,1,1.5,3,3,2" "1.5,1.5,1.5,3,3,2" "1,1,1,2.5,2.5,2.5"; do
  echo "weights $w"; RSX_SWEEP_WEIGHTS=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
echo "sort in fwd0 (old form)"; RSX_SORT_IN_GATHER=0 timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
