cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_adam_window.py tests/test_gpu_dp.py -x -q 2>&1 | tail -8
for w in 1 8; do
RSX_ADAM_WINDOW=$w RSX_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm dp world1 rccl window=$w', d['ms_per_step'], d['value'])"
done
for n in 2 8; do
for w in 1 8; do
RSX_ADAM_WINDOW=$w timeout 600 python bench.py --emulate_world $n --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepfm emulate_world $n window=$w', d['ms_per_step'])"
done
done
