cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-200
