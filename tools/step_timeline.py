#!/usr/bin/env python
"""Every dispatch of the LAST bench step of a rocprofv3 kernel trace (rocpd SQLite) that lasts >= min_us, in start order, with
the gap since the previous dispatch's end: what fills the step beside the named kernels.
    python tools/step_timeline.py <results.db> [min_us]"""
import sqlite3
import sys


def main(path, min_us=50.0):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    last = max(i for i, r in enumerate(rows) if "disp_bracket_kernel" in r[0])     # the step's first kernel
    t0 = rows[last][1]
    tot = 0.0
    for n, s, e in rows[last:]:
        d = (e - s) / 1e3
        tot += d
        if d >= min_us:
            print(f"+{(s - t0) / 1e3:10.1f} us  {d:10.1f} us  {n[:110]}")
    print(f"step span {(rows[-1][2] - t0) / 1e6:.2f} ms, sum of dispatch durations {tot / 1e3:.2f} ms, dispatches {len(rows) - last}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 50.0)
