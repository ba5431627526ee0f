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
    print(f"# kernel-trace --stats summary - {db}")
    print(f"{'kernel':90s} {'calls':>7s} {'total us':>12s} {'avg us':>10s} {'%':>7s}")
    for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
        print(f"{short(name):90s} {calls:7d} {total:12.1f} {avg:10.2f} {pct:7.2f}")


if __name__ == "__main__":
    main()
