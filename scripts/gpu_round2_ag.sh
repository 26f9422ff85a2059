cd $GRAFT_REPO_ROOT
for w in 16 32 64 128 256 512; do
RSX_WIN_SIDE=1 RSX_WIN_WGS=$w timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm side=1 wgs=$w', d['ms_per_step'], d['value'])"
done
RSX_WIN_SIDE=0 timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm side=0', d['ms_per_step'], d['value'])"
RSX_WIN_SIDE=1 RSX_WIN_WGS=64 scripts/prof.sh r02_ag_deepfm_side64 --steps 400 --warmup 48 --no_cpu_baseline > /dev/null
head -12 gpurun_out/r02_ag_deepfm_side64.txt | cut -c1-125
