import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
first = min(i for i, r in enumerate(rows) if "disp_bracket_kernel" in r[0])
t0 = rows[first][1]
for n, s, e in rows[first:]:
    d = (e - s) / 1e3
    if d >= 2000 and (s - t0) / 1e6 < 2500:
        print(f"+{(s - t0) / 1e3:10.1f} us  {d:10.1f} us  {n[:90]}")
