#!/usr/bin/env python3
"""Kernel statistics (calls, total / average / min / max duration) from a rocprofv3 rocpd
database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 bench.py ...` writes
DIR/NAME_results.db on this ROCm).  Prints CSV like rocprofv3's kernel_stats.csv.
usage: rocpd_stats.py <results.db> [> profiles/rNN_kernel_stats_<what>.csv]"""
import re
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), "
                     f"max(end - start) from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, calls, tot, avg, mn, mx in rows:
        n = re.sub(r"\(.*$", "", n)
        print(f'"{n}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}')


if __name__ == "__main__":
    main(sys.argv[1])
