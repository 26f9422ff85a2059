set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/r02_b_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r02_b_bench_driver_cmd.log
timeout 300 python scripts/ingest_bench.py 400000 256 2>&1 | tee gpurun_out/r02_b_ingest_bench.log
