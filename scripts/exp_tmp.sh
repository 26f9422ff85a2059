cd $GRAFT_REPO_ROOT
b() { python bench.py --no_cpu_baseline --no_configs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('launches_per_step'), d['config']['timed_repeats_ms_per_step'])"; }
echo "din fused mlp"; b --model din
echo "din layerwise"; RSX_MLP_FUSE=0 b --model din
timeout 900 python -m pytest tests/test_gpu_din.py tests/test_gpu_knobs.py tests/test_gpu_dp.py -x -q 2>&1 | tail -4
scripts/gpu.sh prof din_mlp --model din --no_cpu_baseline --no_configs --steps 160 --warmup 16 > /dev/null; head -30 gpurun_out/din_mlp.txt | cut -c1-130
