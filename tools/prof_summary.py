#!/usr/bin/env python
"""Summarise a rocprofv3 results database (``*_results.db``, rocpd sqlite) into the small text
files kept under profiles/:   python tools/prof_summary.py <results.db> [--pmc] > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if len(name) > 90:
        name = name[:87] + "..."
    return name


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    if "--pmc" in sys.argv:
        print(f"# PMC counters per kernel (avg per dispatch) - {db}")
        q = ("select kernel_name, counter_name, avg(value), count(*), avg(end-start) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        print(f"{'kernel':90s} {'counter':28s} {'avg value':>16s} {'n':>5s} {'avg ns':>12s}")
        for k, c, v, n, d in cur.execute(q):
            print(f"{short(k):90s} {c:28s} {v:16.1f} {n:5d} {d:12.0f}")
        return
    if "--by-grid" in sys.argv:  # one line per (kernel, grid): the same kernel on different layer shapes
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
        sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
        q = (f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, count(*), avg(d.end - d.start), sum(d.end - d.start) "
             f"from {disp} d join {sym} s on d.kernel_id = s.id group by 1, 2, 3, 4 order by 7 desc limit 60")
        print(f"# per (kernel, grid) - {db}")
        for k, gx, gy, gz, n, avg, tot in cur.execute(q):
            print(f"{short(k)[:70]:70s} grid {gx:8d} {gy:5d} {gz:4d} calls {n:5d} avg {avg / 1e3:8.1f} us total {tot / 1e3:10.1f} us")
        return
    if "--timeline" in sys.argv:  # the last K dispatches in start order: duration and the idle gap before each
        k = int(sys.argv[sys.argv.index("--timeline") + 1])
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
        sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
        q = (f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.start, d.end "
             f"from {disp} d join {sym} s on d.kernel_id = s.id order by d.start desc limit {k + 1}")
        rows = list(cur.execute(q))[::-1]
        print(f"# last {k} dispatches - {db}")
        busy = 0
        for prev, r in zip(rows, rows[1:]):
            busy += r[6] - r[5]
            print(f"{short(r[0])[:64]:64s} wgs {r[1] * r[2] * r[3] // max(r[4], 1):6d} x {r[4]:4d}  {(r[6] - r[5]) / 1e3:8.2f} us  gap {(r[5] - prev[6]) / 1e3:7.2f} us")
        print(f"# span {(rows[-1][6] - rows[1][5]) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
        return
    print(f"# kernel-trace --stats summary - {db}")
    print(f"{'kernel':90s} {'calls':>7s} {'total us':>12s} {'avg us':>10s} {'%':>7s}")
    for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
        print(f"{short(name):90s} {calls:7d} {total:12.1f} {avg:10.2f} {pct:7.2f}")


if __name__ == "__main__":
    main()
