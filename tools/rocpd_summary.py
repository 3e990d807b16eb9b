#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace --stats) per kernel:
calls, total / average / min / max duration.

usage: rocpd_summary.py results.db out.csv [out.txt]     write the CSV and (round 6) a human-readable .txt made FROM THE SAME ROWS
       rocpd_summary.py --txt-from-csv in.csv out.txt    regenerate a .txt from a committed CSV
       rocpd_summary.py --check in.txt in.csv            exit 1 unless the .txt's header (total ms, dispatches) and its rows are those
                                                         of the CSV (a two-stream header sat over one-stream rows)
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void\s+", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def write_txt(rows, path, source):
    """rows: the CSV's rows incl. the header row"""
    body = rows[1:]
    total = sum(float(r[2]) for r in body)
    calls = sum(int(r[1]) for r in body)
    with open(path, "w") as fh:
        fh.write(f"# total kernel time {total:.1f} ms over {calls} dispatches\n# rows: {source} (written by tools/rocpd_summary.py from those rows)\n")
        fh.write(f"{'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}  kernel\n")
        for k, c, t, a, mn, mx, pct in body:
            fh.write(f"{int(c):7d} {float(t):10.3f} {float(a):10.2f} {float(pct):6.2f}  {k}\n")


def check(txt, csv_path):
    rows = list(csv.reader(open(csv_path)))[1:]
    total, calls = sum(float(r[2]) for r in rows), sum(int(r[1]) for r in rows)
    lines = open(txt).read().splitlines()
    m = re.match(r"# total kernel time ([0-9.]+) ms over (\d+) dispatches", lines[0]) if lines else None
    if not m:
        sys.exit(f"{txt}: no header line")
    if abs(float(m.group(1)) - total) > 0.06 or int(m.group(2)) != calls:
        sys.exit(f"{txt}: header says {m.group(1)} ms / {m.group(2)} dispatches, {csv_path} holds {total:.1f} ms / {calls}")
    data = [ln.split(None, 4) for ln in lines if ln and ln[0] != "#" and ln.split()[0].isdigit()]
    for (c, t, a, pct, k), r in zip(data, rows):
        if int(c) != int(r[1]) or abs(float(t) - float(r[2])) > 0.002 or k.strip() != r[0]:
            sys.exit(f"{txt}: row '{k.strip()[:60]}' ({c} calls, {t} ms) is not the CSV's ({r[1]} calls, {r[2]} ms)")
    if len(data) != len(rows):
        sys.exit(f"{txt}: {len(data)} rows, {csv_path}: {len(rows)}")
    print(f"{txt}: consistent with {csv_path} ({total:.1f} ms, {calls} dispatches, {len(rows)} kernels)")


def main():
    if sys.argv[1] == "--check":
        return check(sys.argv[2], sys.argv[3])
    if sys.argv[1] == "--txt-from-csv":
        return write_txt(list(csv.reader(open(sys.argv[2]))), sys.argv[3], sys.argv[2].split("/")[-1])
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
    if len(sys.argv) > 3:
        write_txt([tuple(str(x) for x in r) for r in out], sys.argv[3], sys.argv[2].split("/")[-1])
    print(f"# total kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches", file=sys.stderr)


if __name__ == "__main__":
    main()
