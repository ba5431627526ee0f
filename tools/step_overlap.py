#!/usr/bin/env python
"""Per-step occupancy of a training run from a rocprofv3 kernel trace (rocpd sqlite): python tools/step_overlap.py <results.db> [marker]

Steps are cut at the marker kernel (default ``pack_conv_batch_kernel``: once per step).  For the last steps: span (first start to
last end), the union of the busy intervals (any kernel running), idle = span - union, the sum of the durations (> union where two
streams / graph branches overlap) and the same per queue.  Eager and captured-graph runs of one workload are compared with it."""
import sqlite3
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    db = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "pack_conv_batch_kernel"
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
    cuts = [i for i, r in enumerate(rows) if marker in r[0]]
    print(f"# {db}: {len(rows)} dispatches, {len(cuts)} steps cut at {marker}; queue column: {qcol}")
    steps = [rows[a:b] for a, b in zip(cuts, cuts[1:])][-8:]
    for st in steps:
        span = max(r[2] for r in st) - st[0][1]
        busy = union([(r[1], r[2]) for r in st])
        total = sum(r[2] - r[1] for r in st)
        perq = defaultdict(list)
        for r in st:
            perq[r[3]].append((r[1], r[2]))
        q = ", ".join(f"q{k}: {len(v)} launches {union(v) / 1e3:.0f} us" for k, v in sorted(perq.items(), key=lambda kv: -len(kv[1])))
        print(f"launches {len(st):4d}  span {span / 1e3:8.1f} us  busy {busy / 1e3:8.1f}  idle {(span - busy) / 1e3:7.1f}  sum {total / 1e3:8.1f}  | {q}")


if __name__ == "__main__":
    main()
