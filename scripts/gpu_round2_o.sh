cd $GRAFT_REPO_ROOT
for m in deepfm fm dcn xdeepfm din; do
  timeout 300 python bench.py --model $m --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r02_o_all.log
done
RSX_STAGE_A_FINAL=0 timeout 300 python bench.py --model dcn --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dcn stageA-final off', d['ms_per_step'])" | tee -a gpurun_out/r02_o_all.log
RSX_TOWER_DXG=0 timeout 300 python bench.py --model dcn --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dcn dxg off', d['ms_per_step'])" | tee -a gpurun_out/r02_o_all.log
timeout 300 python bench.py --model xdeepfm --cin_bf16 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm bf16', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r02_o_all.log
scripts/prof.sh r02_o_dcn_kernel_stats --model dcn --steps 400 --warmup 50 --no_cpu_baseline > /dev/null
head -16 gpurun_out/r02_o_dcn_kernel_stats.txt | cut -c1-130
