cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_adam_window.py -x -q 2>&1 | tail -5
for w in 1 2 3 4; do
RSX_ADAM_WINDOW=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm window=$w', d['ms_per_step'], d['value'])"
done
scripts/prof.sh r02_w_deepfm_window4_kernel_stats --steps 400 --warmup 50 --no_cpu_baseline > /dev/null
head -16 gpurun_out/r02_w_deepfm_window4_kernel_stats.txt | cut -c1-130
