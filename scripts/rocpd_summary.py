#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as the text table committed under profiles/.

usage: python scripts/rocpd_summary.py gpurun_out/<dir>/<name>_results.db [title] > profiles/rNN_<what>.txt
(rocprofv3 7.2 writes rocpd .db files; this is the `--kernel-trace --stats` per-kernel table.)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    tot = c.execute("select sum(end-start) from kernels").fetchone()[0] or 1
    n = c.execute("select count(*) from kernels").fetchone()[0]
    print("# %s" % title)
    print("# source: rocprofv3 --kernel-trace --stats (rocpd db), %d dispatches, %.3f ms total kernel time" % (n, tot / 1e6))
    print("%-72s %8s %10s %10s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "pct"))
    q = ("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
         "group by name order by sum(end-start) desc")
    for name, cnt, avg, mn, mx, sm in c.execute(q):
        print("%-72s %8d %10.2f %10.2f %10.2f %10.3f %6.2f" % (name[:72], cnt, avg / 1e3, mn / 1e3, mx / 1e3, sm / 1e6, 100.0 * sm / tot))


if __name__ == "__main__":
    main()
