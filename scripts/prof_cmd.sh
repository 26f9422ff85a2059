#!/bin/bash
# usage (GPU box): scripts/prof_cmd.sh <name> <python script + args, relative to the repo root>
# rocprofv3 --kernel-trace --stats of an arbitrary command -> gpurun_out/<name>.txt (per-kernel table)
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o r -- python $root/"$@" > $root/gpurun_out/$name.log 2>&1
db=$(ls /tmp/prof_$name/*.db | head -1)
python $root/scripts/rocpd_summary.py $db "rocprofv3 --kernel-trace --stats -- python $*" > $root/gpurun_out/$name.txt
head -12 $root/gpurun_out/$name.txt | cut -c1-140
rm -rf /tmp/prof_$name
