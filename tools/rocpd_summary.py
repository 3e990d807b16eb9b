#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace --stats) per kernel:
calls, total / average / min / max duration.  usage: rocpd_summary.py results.db [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void\s+", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {namec}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {namec} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows)
    out = [("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    for n, c, t, a, mn, mx in rows:
        out.append((short(n), c, f"{t / 1e6:.3f}", f"{a / 1e3:.2f}", f"{mn / 1e3:.2f}", f"{mx / 1e3:.2f}", f"{100.0 * t / total:.2f}"))
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)
    print(f"# total kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches", file=sys.stderr)


if __name__ == "__main__":
    main()
