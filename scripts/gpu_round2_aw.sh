# GPU box: per-rank COMPUTE of an N-rank data-parallel DeepFM step with the collectives emulated on one GPU (bench.py --emulate_world)
cd $GRAFT_REPO_ROOT
for n in 2 4 8; do
  timeout 300 python bench.py --emulate_world $n --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('emulate_world $n: ms_per_step %.5f  adam_window %s' % (d['ms_per_step'], d['config']['adam_window']))"
done
