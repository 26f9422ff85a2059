cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_din.py -x -q 2>&1 | tail -3
scripts/gpu.sh prof din_mlp3 --model din --no_cpu_baseline --no_configs --steps 160 --warmup 16 > /dev/null; grep -E "pool|finish|calls" gpurun_out/din_mlp3.txt | head -5 | cut -c1-130
python bench.py --model din --no_cpu_baseline --no_configs 2>/dev/null | tail -1 | cut -c1-150
