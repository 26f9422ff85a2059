# GPU box: steps per optimizer window above batch 1024 (dcn.py bs 4096), A/B
cd $GRAFT_REPO_ROOT
b() { timeout 300 python bench.py --model dcn --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['config']['adam_window'])"; }
for k in 4 8 6 4 8; do RSX_ADAM_WINDOW_LARGE=$k b window_large=$k; done
