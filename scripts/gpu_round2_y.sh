cd $GRAFT_REPO_ROOT
scripts/prof.sh r02_y_dcn_window4_kernel_stats --model dcn --steps 200 --warmup 32 --no_cpu_baseline > /dev/null
head -18 gpurun_out/r02_y_dcn_window4_kernel_stats.txt | cut -c1-130
RSX_ADAM_WINDOW=1 scripts/prof.sh r02_y_dcn_window1_kernel_stats --model dcn --steps 200 --warmup 32 --no_cpu_baseline > /dev/null
head -16 gpurun_out/r02_y_dcn_window1_kernel_stats.txt | cut -c1-130
