cd $GRAFT_REPO_ROOT
python scripts/window_sweep_time.py 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_adam_window.py tests/test_gpu_embedding.py -x -q 2>&1 | tail -5
for w in 1 4; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm window=$w', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done
for w in 1 2 4; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --model dcn --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dcn window=$w', d['ms_per_step'], d['value'])"
done
scripts/prof.sh r02_z_dcn_window4_kernel_stats --model dcn --steps 200 --warmup 32 --no_cpu_baseline > /dev/null
head -8 gpurun_out/r02_z_dcn_window4_kernel_stats.txt | cut -c1-130
