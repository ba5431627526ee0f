#!/usr/bin/env python
"""HBM traffic of the conv kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).

    python tools/pmc_traffic.py <fetch_results.db> <write_results.db> [kernel substring[,substring..]] [key=value ...] > profiles/..json

``key=value`` pairs (cfg, size, batch, workload, dtype, date) are stored beside the numbers, together with the hash of the
kernel sources (``bench.kernel_generation``): bench.py reports the measurement only for the run it was taken on.

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB
(bytes = value * 1024); on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced read, so it
is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import json
import sqlite3
import sys


def per_launch(db, counter, match):
    cur = sqlite3.connect(db).cursor()
    tot, n = 0.0, 0
    for m in match.split(","):
        q = ("select sum(value), count(*) from counters_collection where counter_name = ? and kernel_name like ?")
        t1, n1 = cur.execute(q, (counter, f"%{m}%")).fetchone()
        tot, n = tot + (t1 or 0.0), n + (n1 or 0)
    return tot, n


def main():
    fetch_db, write_db = sys.argv[1], sys.argv[2]
    match = sys.argv[3] if len(sys.argv) > 3 else "conv_igemm"
    keys = dict(kv.split("=", 1) for kv in sys.argv[4:])
    for k in ("size", "batch"):
        if k in keys:
            keys[k] = int(keys[k])
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench
    keys["kernel_generation"] = bench.kernel_generation(half=keys.get("dtype", "f32") != "f32")
    last = int(keys.pop("last_forward", 0))
    if last:
        # only the conv launches of the LAST whole detector forward of each pass (tools/pmc_layers.py's rule: behind the last stem
        # launch; slab-reduce launches charged to their conv launch) - the plain average below also counts whatever else the process
        # launched with a matching name (autotuning candidates, the measured pair-vs-one-launch choice), which inflated the 16-bit figure
        import pmc_layers
        fl = pmc_layers.launches(fetch_db, "FETCH_SIZE", last)
        wl = pmc_layers.launches(write_db, "WRITE_SIZE", last)
        f, nf, w, nw = sum(v for _n, v, _d in fl), len(fl), sum(v for _n, v, _d in wl), len(wl)
        match = f"last forward: {last} conv launches (+ their slab-reduce launches)"
    else:
        f, nf = per_launch(fetch_db, "FETCH_SIZE", match)
        w, nw = per_launch(write_db, "WRITE_SIZE", match)
    fetch_bytes = 2.0 * f * 1024.0 / max(nf, 1)
    write_bytes = w * 1024.0 / max(nw, 1)
    print(json.dumps({
        **keys,
        "kernel_match": match, "launches_fetch_pass": nf, "launches_write_pass": nw,
        "fetch_bytes_per_launch": round(fetch_bytes), "write_bytes_per_launch": round(write_bytes),
        "hbm_bytes_per_launch": round(fetch_bytes + write_bytes),
        "note": "FETCH_SIZE x2 (gfx950 half-count of wide reads), KiB -> bytes; WRITE_SIZE uncalibrated; "
                "average over every matching launch of `python bench.py --no-batch-sweep --no-accuracy --no-bf16-line [--dtype bf16] --steps 3 --warmup 1`",
    }))


if __name__ == "__main__":
    main()
