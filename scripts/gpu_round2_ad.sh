cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_embedding.py -x -q -k overlapped_optimizer 2>&1 | grep -E "assert|Error|diff|tables|w1|dense" | head -10
for w in 2 3 4 5 6 7 8; do
RSX_ADAM_WINDOW=$w timeout 600 python -m pytest tests/test_gpu_embedding.py -x -q -k overlapped_optimizer 2>&1 | tail -1 | sed "s/^/window=$w: /"
done
