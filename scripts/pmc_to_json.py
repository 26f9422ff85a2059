#!/usr/bin/env python
"""profiles/<tag>_pmc_FETCH_SIZE_deepfm.txt + <tag>_pmc_WRITE_SIZE_deepfm.txt (scripts/pmc.sh, separate --pmc passes) ->
profiles/pmc_adam_window_k.json and profiles/pmc_adam_multi_k.json, the files bench.py's `roofline.traffic` reads.
usage: python scripts/pmc_to_json.py r06_z 6"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], int(sys.argv[2])


def table(name):
    out = {}
    for line in open(os.path.join(ROOT, "profiles", "%s_pmc_%s_deepfm.txt" % (tag, name))):
        m = re.match(r"^(.*?)\s+%s\s+calls=\s*(\d+)\s+mean=([0-9.]+)" % name, line)
        if m and int(m.group(2)) > 4:
            out.setdefault(m.group(1).strip(), float(m.group(3)))
    return out


fetch, write = table("FETCH_SIZE"), table("WRITE_SIZE")
R, D = 840646, 16
n_sparse = R * D + R                    # table elements + the first-order vector
n_dense = 72903 + 1                     # deepfm.py's dense arena (624x100 + 100x100 + ... ; bench.py prints the exact figure)
for kern, key, alg in (("adam_window_k<7>", "adam_window_k", 24 * n_sparse + 4 * 8 * R),
                       ("adam_multi_k", "adam_multi_k", 24 * n_sparse + 32 * n_dense)):
    fk = next((k for k in fetch if k.startswith("void " + kern) or k.startswith(kern)), None)
    wk = next((k for k in write if k.startswith("void " + kern) or k.startswith(kern)), None)
    if fk is None or wk is None:
        print("no rows for", kern)
        continue
    f, w = fetch[fk], write[wk]
    out = {"kernel": kern, "model": "deepfm",
           "command": "rocprofv3 --pmc <CTR> --kernel-trace --output-format csv -- python bench.py --no_cpu_baseline --no_configs "
                      "--steps 64 --warmup 32  (one counter per pass: scripts/pmc.sh)",
           "FETCH_SIZE_kb_per_launch": f, "WRITE_SIZE_kb_per_launch": w,
           "corrections": "MI355X_MICROARCH.md section HBM: values are KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide "
                          "coalesced read -> x2; WRITE_SIZE is taken at face value",
           "traffic_bytes_per_launch": int((2 * f + w) * 1024), "algorithmic_bytes_per_launch": alg, "round": rnd,
           "source": "profiles/%s_pmc_FETCH_SIZE_deepfm.txt, profiles/%s_pmc_WRITE_SIZE_deepfm.txt" % (tag, tag)}
    if key == "adam_window_k":
        out["window_steps"] = 8
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_%s.json" % key), "w"), indent=1)
    print(key, "traffic %.1f MB, algorithmic %.1f MB, ratio %.3f" % (out["traffic_bytes_per_launch"] / 1e6, alg / 1e6, out["traffic_bytes_per_launch"] / alg))
