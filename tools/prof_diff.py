#!/usr/bin/env python
"""Per-step kernel table from TWO rocprofv3 runs of the same command with different step counts: whatever the process does
once (engine planning, autotuning, first-use packing, the accuracy pass) cancels in the difference.
usage: python tools/prof_diff.py <short_results.db> <long_results.db> <steps_long - steps_short | kernel-name-substring>
(a kernel name instead of a number: the step count is the difference of that kernel's launch counts - one launch per step)"""
import sqlite3
import sys

from prof_summary import short


def table(db):
    cur = sqlite3.connect(db).cursor()
    return {name: (calls, total) for name, calls, total, _avg, _pct in cur.execute("select * from top_kernels")}


def main():
    a, b = table(sys.argv[1]), table(sys.argv[2])
    try:
        n = float(sys.argv[3])
    except ValueError:
        n = float(sum(c for k, (c, _) in b.items() if sys.argv[3] in k) - sum(c for k, (c, _) in a.items() if sys.argv[3] in k))
    rows = []
    for k, (calls_b, total_b) in b.items():
        calls_a, total_a = a.get(k, (0, 0.0))
        rows.append(((total_b - total_a) / n, (calls_b - calls_a) / n, k))
    rows.sort(reverse=True)
    whole = sum(r[0] for r in rows)
    print(f"# per-step kernel time = ({sys.argv[2]} - {sys.argv[1]}) / {n:g} steps; sum {whole / 1e3:.2f} ms")
    print(f"{'kernel':90s} {'calls/step':>10s} {'us/step':>10s} {'%':>6s}")
    for us, calls, k in rows:
        if us > 0.0005 * whole:
            print(f"{short(k):90s} {calls:10.1f} {us:10.1f} {100 * us / whole:6.2f}")


if __name__ == "__main__":
    main()
