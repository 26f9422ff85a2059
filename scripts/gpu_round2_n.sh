cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cin_bf16.py -q -k "sweep_carriers" 2>&1 | grep -v amdgpu.ids | grep -B2 -A12 "AssertionError\|assert torch.equal\|passed\|failed" | cut -c1-300 | tail -60
