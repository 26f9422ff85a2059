# GPU box: bench.py as 2 data-parallel ranks on ONE GPU (gloo collectives, both ranks on cuda:0): a smoke test of the N > 1
# code path of bench.py / dist.py, not a measurement.  Repeated: the rendezvous + first collectives must not hang.
cd $GRAFT_REPO_ROOT
T=${T:-90}
for rep in 1 2 3 4; do
for r in 0 1; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29611 + rep)) RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 RSX_DIST_BACKEND=gloo PYTHONFAULTHANDLER=1 \
    timeout -s ABRT $T python -X faulthandler bench.py --gpus 2 --steps 20 --warmup 5 --no_cpu_baseline $EXTRA > gpurun_out/dp2_rank${r}_$rep.log 2>&1 &
done
wait
echo "rep $rep: $(tail -1 gpurun_out/dp2_rank0_$rep.log | cut -c1-160)"
grep -n "File \"/root/repo\|File \"$PWD" gpurun_out/dp2_rank0_$rep.log | head -12
grep -n "File \"/root/repo\|File \"$PWD" gpurun_out/dp2_rank1_$rep.log | head -12
done
