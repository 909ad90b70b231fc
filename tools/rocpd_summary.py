#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd/SQLite) kernel trace into the classic --stats table.

    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_results.db > profiles/r1_bench_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{short}` | {n} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.2f} |")
    extra = [c_ for c_ in ("vgpr_count", "sgpr_count", "lds_size", "workgroup_size", "grid_size") if c_ in cols]
    if extra:
        print("\n| kernel | " + " | ".join(extra) + " |")
        print("|---|" + "---:|" * len(extra))
        for r in c.execute(f"select name, {', '.join('max(' + e + ')' for e in extra)} from kernels group by name order by sum(end-start) desc"):
            print(f"| `{r[0][:80]}` | " + " | ".join(str(v) for v in r[1:]) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
