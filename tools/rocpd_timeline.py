#!/usr/bin/env python
"""Timeline excerpt of a rocprofv3 kernel trace (rocpd SQLite): the dispatches around the longest instances of a kernel.
    python tools/rocpd_timeline.py <results.db> <name substring> [context]"""
import sqlite3
import sys


def main(path, pat, ctx=3):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    hits = sorted((i for i, r in enumerate(rows) if pat in r[0]), key=lambda i: rows[i][1] - rows[i][2])[:3]
    for i in sorted(hits):
        print("----")
        for j in range(max(0, i - ctx), min(len(rows), i + ctx + 1)):
            n, s, e = rows[j]
            print(f"{'>>' if j == i else '  '} +{(s - rows[i][1]) / 1e3:10.1f} us  {(e - s) / 1e3:10.1f} us  {n[:100]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
