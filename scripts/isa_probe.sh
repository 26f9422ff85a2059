#!/bin/bash
# usage: scripts/isa_probe.sh <file.hip> <mangled-kernel-name-substring> [full]
# Compiles one translation unit with -DRSX_ISA_PROBE (riders compiled out) and prints, for the kernel whose mangled name
# contains the substring, the register / scratch figures and the skeleton of its instruction stream (memory operations,
# waits, barriers, MFMAs, branches) -- what decides the length of a latency-bound kernel's dependent chain.
root=$(cd "$(dirname "$0")/.." && pwd)
src=$1; pat=$2; full=$3
out=/tmp/isa_probe_$$; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DRSX_ISA_PROBE -I$root/include -I$root/recsys_amd/csrc \
  -c $src -o $out/x.o -save-temps=obj 2>&1 | grep -E "error" 
python3 - $out/$(basename ${src%.*})-hip-amdgcn-amd-amdhsa-gfx950.s "$pat" "$full" <<'PY'
import re, sys, collections
s = open(sys.argv[1]).read()
pat, full = sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", s, re.M) if pat in m.group(1)]
for name in names:
    i = s.index(name + ":"); j = s.index(".end_amdhsa_kernel", i)
    meta = s[s.index(".amdhsa_kernel " + name):j]
    lines = [l.strip() for l in s[i:s.index(".amdhsa_kernel " + name)].split("\n")]
    lines = [l for l in lines if l and not l.startswith(";") and not l.startswith(".") ]
    ins = [l for l in lines if not l.endswith(":")]
    g = lambda k: re.search(k + r"\s+(\d+)", meta).group(1)
    print("==", name, "instructions", len(ins), "vgpr", g("next_free_vgpr"), "scratch", g("private_segment_fixed_size"))
    print("  ", ", ".join("%s %d" % kv for kv in collections.Counter(l.split()[0] for l in ins).most_common(18)))
    if full:
        out = []
        for k, l in enumerate(lines):
            op = l.split()[0]
            if l.endswith(":"):
                out.append("\n" + l)
            elif op.startswith(("global_", "buffer_", "s_waitcnt", "s_barrier", "v_mfma", "ds_", "scratch", "s_cbranch", "s_branch", "s_endpgm", "s_load")):
                out.append("%d:%s" % (k, l.split(";")[0].strip() if op.startswith(("s_waitcnt", "s_cbranch", "s_branch")) else " ".join(l.replace(",", " ").split()[:2])))
        print(" | ".join(out))
PY
rm -rf $out
