cd $GRAFT_REPO_ROOT
RSX_WINDOW_SETS=2 timeout 200 python scripts/e2e_host_timeline.py 2>&1 | grep -v amdgpu | tail -2
timeout 200 python scripts/e2e_host_timeline.py 2>&1 | grep -v amdgpu | tail -2

