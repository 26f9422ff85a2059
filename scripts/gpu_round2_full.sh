cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02_full_tests.log
tail -5 gpurun_out/r02_full_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r02_full_bench_20.json
cat gpurun_out/r02_full_bench_20.json | cut -c1-400
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r02_full_bench_default.json
cat gpurun_out/r02_full_bench_default.json | cut -c1-300
