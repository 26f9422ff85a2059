#!/bin/bash
# usage (GPU box): scripts/pmc.sh <name> <counter> <bench args...>   -- one counter group per pass (TCC has 4 slots)
name=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o r -- python $root/bench.py "$@" > $root/gpurun_out/pmc_$name.log 2>&1
f=$(ls /tmp/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)
python - "$f" "$ctr" > $root/gpurun_out/pmc_$name.txt <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = (row["Kernel_Name"][:60], row["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(row["Counter_Value"])
print("# rocprofv3 --pmc %s --kernel-trace: per-kernel mean counter value per dispatch" % ctr)
for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(__import__("os").environ.get("PMC_ROWS", "12"))]:
    print("%-62s %-12s calls=%6d mean=%.1f" % (k, c, n, s / n))
PY
cat $root/gpurun_out/pmc_$name.txt | head -8
rm -rf /tmp/pmc_$name
