#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/prof.sh <name> <bench args...>
# Runs rocprofv3 --kernel-trace --stats on bench.py, writes the per-kernel table to gpurun_out/<name>.txt and deletes
# the (large) rocpd database so gpurun_out/ stays under the 64 MiB pull limit.
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o r -- python $root/bench.py "$@" > $root/gpurun_out/$name.log 2>&1
db=$(ls /tmp/prof_$name/*.db | head -1)
python $root/scripts/rocpd_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py $*" > $root/gpurun_out/$name.txt
python - "$db" >> $root/gpurun_out/$name.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
mid = len(rows) // 2
print("\n# timeline sample (us from first row; one training step ~ between two field_sort_k):")
for r in rows[mid:mid + 130]:
    print("%-150s start=%10.2f dur=%8.2f" % (r[0][:150], (r[1] - rows[mid][1]) / 1e3, (r[2] - r[1]) / 1e3))
PY
python $root/scripts/occupancy_table.py $db > $root/gpurun_out/${name}_occupancy.txt 2>&1
grep metric $root/gpurun_out/$name.log | cut -c1-300
rm -rf /tmp/prof_$name
