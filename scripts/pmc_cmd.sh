#!/bin/bash
# usage (GPU box): scripts/pmc_cmd.sh <name> "<counter> [<counter> ...]" <python script + args, relative to the repo root>
# One rocprofv3 --pmc pass (with --kernel-trace only: gpurun refuses pmc + other trace domains) of an arbitrary command;
# writes the per-kernel mean counter values per dispatch to gpurun_out/pmc_<name>.txt.
name=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o r -- python $root/"$@" > $root/gpurun_out/pmc_$name.log 2>&1
f=$(ls /tmp/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)
python - "$f" "$ctr" "$*" > $root/gpurun_out/pmc_$name.txt <<'PY'
import csv, sys, collections
f, ctr, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
with open(f) as fh:
    for row in csv.DictReader(fh):
        a = agg[row["Kernel_Name"][:70]][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
print("# rocprofv3 --pmc %s --kernel-trace -- python %s" % (ctr, cmd))
print("# per-kernel MEAN counter value per dispatch (FETCH_SIZE / WRITE_SIZE in KB; gfx950: double FETCH_SIZE for wide streaming reads)")
names = ctr.split()
print("%-72s %7s " % ("kernel", "calls") + " ".join("%20s" % n for n in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:16]:
    calls = max(v[0] for v in d.values())
    print("%-72s %7d " % (k, calls) + " ".join("%20.1f" % (d[n][1] / d[n][0] if n in d and d[n][0] else float("nan")) for n in names))
PY
head -12 $root/gpurun_out/pmc_$name.txt
rm -rf /tmp/pmc_$name
