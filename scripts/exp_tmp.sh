cd $GRAFT_REPO_ROOT
scripts/gpu.sh test 2>&1 | tail -4
scripts/round_end.sh r04_z 2>&1 | tail -24
( echo "# din.py bs 1024: A/B on one box (bench.py --model din --no_cpu_baseline --no_configs; ms per step, launches per step)"
  b() { python bench.py --model din --no_cpu_baseline --no_configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'launches/step', d['config'].get('launches_per_step'), d['config']['timed_repeats_ms_per_step'])"; }
  echo -n "default: "; b
  echo -n "RSX_SCATTER_RIDERS=0 (the attention weight-gradient reduces inside the finish launch): "; RSX_SCATTER_RIDERS=0 b
  echo -n "RSX_DIN_GATHER_RIDE=0 (the lookups as their own launch): "; RSX_DIN_GATHER_RIDE=0 b
  echo -n "RSX_MLP_REDUCE_RIDE=0 (mlp_layer's weight-gradient reduce as its own launch): "; RSX_MLP_REDUCE_RIDE=0 b
  echo -n "RSX_MLP_REDUCE_RIDE=0 RSX_MLP_REDUCE_SIDE=1 (.. on the side stream): "; RSX_MLP_REDUCE_RIDE=0 RSX_MLP_REDUCE_SIDE=1 b
  echo -n "RSX_MLP_FUSE=0 (mlp_layer as 8 launches): "; RSX_MLP_FUSE=0 b
  echo -n "RSX_DIN_SIDE_SORT=0 (sort + sweep in line): "; RSX_DIN_SIDE_SORT=0 b
  echo -n "all five off (the step as of round 3 + the faster attention backward + the prepare fix): "; RSX_SCATTER_RIDERS=0 RSX_DIN_GATHER_RIDE=0 RSX_MLP_REDUCE_RIDE=0 RSX_MLP_FUSE=0 RSX_DIN_SIDE_SORT=0 b
) > gpurun_out/r04_i_din_ab.txt 2>&1
( echo "# dcn.py bs 4096, A/B on one box: ms per step, launches per step"
  b() { python bench.py --model dcn --no_cpu_baseline --no_configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'launches/step', d['config'].get('launches_per_step'), d['config']['timed_repeats_ms_per_step'])"; }
  for i in 1 2; do echo -n "default: "; b; echo -n "RSX_SCATTER_RIDERS=0 (the dW / cross-gradient reduces as their own launches): "; RSX_SCATTER_RIDERS=0 b; echo -n "RSX_GATHER_CROSS=0 (lookup and cross forward as two launches): "; RSX_GATHER_CROSS=0 b; echo -n "both off: "; RSX_SCATTER_RIDERS=0 RSX_GATHER_CROSS=0 b; done
) > gpurun_out/r04_k_dcn_riders_ab.txt 2>&1
cat gpurun_out/r04_i_din_ab.txt gpurun_out/r04_k_dcn_riders_ab.txt
