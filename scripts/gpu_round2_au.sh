# GPU box: the unprofiled bench line of every BASELINE config, one after the other.
cd $GRAFT_REPO_ROOT
b() { timeout 300 python bench.py "$@" --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s ms_per_step %.5f  value %.0f' % ('$*', d['ms_per_step'], d['value']))"; }
b --model deepfm
b --model fm
b --model dcn
b --model xdeepfm
b --model xdeepfm --cin_bf16
b --model din
b --model fm --host_input --steps 800 --warmup 96
