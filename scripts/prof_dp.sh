#!/bin/bash
# GPU box: kernel timeline of the data-parallel step through RCCL at world 1 (segmented graphs + eager collectives)
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
RSX_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_dp -o r -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 $root/bench.py --gpus 1 --steps 300 --warmup 60 --no_cpu_baseline "$@" > $root/gpurun_out/dp1.log 2>&1
db=$(ls /tmp/prof_dp/*/*.db /tmp/prof_dp/*.db 2>/dev/null | head -1)
python - "$db" > $root/gpurun_out/dp1.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
mid = len(rows) // 2
print("# timeline sample (us): world-1 RCCL data-parallel step")
for r in rows[mid:mid + 60]:
    print("%-90s start=%10.2f dur=%8.2f" % (r[0][:90], (r[1] - rows[mid][1]) / 1e3, (r[2] - r[1]) / 1e3))
PY
grep metric $root/gpurun_out/dp1.log | cut -c1-200
rm -rf /tmp/prof_dp
