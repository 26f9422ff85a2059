#!/bin/bash
# usage (GPU box): scripts/pmc_table.sh <name> "<counters>" <bench args...>
# one rocprofv3 --pmc pass (kernel-trace only) of bench.py; prints a kernel x counter table of per-dispatch means, and, when
# SQ_WAVES is among the counters, the per-WAVE means (instructions a wave executes: the currency of a latency-bound kernel).
name=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o r -- python $root/bench.py "$@" > $root/gpurun_out/pmc_$name.log 2>&1
f=$(ls /tmp/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)
python - "$f" "$ctr" "$*" > $root/gpurun_out/pmc_$name.txt <<'PY'
import csv, sys, collections
f, ctr, args = sys.argv[1], sys.argv[2].split(), sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"][:56]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == ctr[0]:
            cnt[k] += 1
print("# rocprofv3 --pmc %s --kernel-trace -- python bench.py %s" % (" ".join(ctr), args))
print("# per-dispatch means" + (" | per-wave means" if "SQ_WAVES" in ctr else ""))
print("%-58s %7s " % ("kernel", "calls") + " ".join("%14s" % c[-14:] for c in ctr))
for k, n in cnt.most_common(24):
    line = "%-58s %7d " % (k, n) + " ".join("%14.1f" % (agg[k][c] / n) for c in ctr)
    if "SQ_WAVES" in ctr and agg[k]["SQ_WAVES"] > 0:
        line += " | " + " ".join("%9.1f" % (agg[k][c] / agg[k]["SQ_WAVES"]) for c in ctr if c != "SQ_WAVES")
    print(line)
PY
head -30 $root/gpurun_out/pmc_$name.txt | cut -c1-250
rm -rf /tmp/pmc_$name
