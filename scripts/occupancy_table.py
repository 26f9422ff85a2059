#!/usr/bin/env python
"""GPU box: per kernel of a rocprofv3 --kernel-trace database, the launch shape against what the chip holds at once --
workgroups, threads, LDS and registers per workgroup -> workgroups per CU -> ROUNDS of the 256 CUs the launch needs.  A launch
that needs 1.1 rounds runs its last few workgroups alone (tower_bwd_big_k 41 -> 35 us).
usage: occupancy_table.py <rocpd .db> [min total ms]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
if len(sys.argv) > 2 and sys.argv[2] == "cols":
    print(cols)
    sys.exit(0)


def col(*names):
    for n in names:
        if n in cols:
            return n
    return None


gx, gy, gz = col("grid_x", "grid_size_x"), col("grid_y", "grid_size_y"), col("grid_z", "grid_size_z")
wx, wy, wz = col("workgroup_x", "workgroup_size_x"), col("workgroup_y", "workgroup_size_y"), col("workgroup_z", "workgroup_size_z")
lds, vg, ag, sg = col("lds_size", "lds_block_size", "group_segment_size"), col("arch_vgpr_count", "vgpr_count"), col("accum_vgpr_count"), col("sgpr_count")
q = "select name, start, end, %s from kernels" % ", ".join(x if x else "0" for x in (gx, gy, gz, wx, wy, wz, lds, vg, ag, sg))
agg = collections.OrderedDict()
for r in c.execute(q):
    name, st, en, x, y, z, a, b, cc, l, v, acc, s = r
    key = (name[:60], x * max(y, 1) * max(z, 1), a * max(b, 1) * max(cc, 1), l, v, acc)
    d = agg.setdefault(key, [0, 0.0])
    d[0] += 1
    d[1] += (en - st) / 1e3
print("# columns of the kernels view used:", gx, wx, lds, vg, ag)
print("%-60s %6s %8s %6s %7s %5s %5s %7s %7s %8s" % ("kernel", "calls", "threads", "block", "lds", "vgpr", "agpr", "wg/CU", "rounds", "avg_us"))
for (name, threads, block, l, v, acc), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    if not block:
        continue
    wgs = threads // block if threads % block == 0 and threads >= block else threads      # (grid given in threads or in workgroups)
    waves = (block + 63) // 64
    regs = ((v or 0) + (acc or 0) + 7) // 8 * 8
    per_simd = 8 if regs == 0 else min(8, 512 // max(regs, 1))
    by_reg = per_simd * 4 // waves if waves <= 4 * per_simd else 0
    by_lds = (160 * 1024) // l if l else 99
    by_waves = 32 // waves
    per_cu = max(1, min(by_reg if by_reg else 1, by_lds, by_waves))
    print("%-60s %6d %8d %6d %7d %5d %5d %7d %7.2f %8.2f" % (name, n, wgs, block, l or 0, v or 0, acc or 0, per_cu, wgs / (256.0 * per_cu), t / n))
