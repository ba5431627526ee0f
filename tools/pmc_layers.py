#!/usr/bin/env python
"""Per-layer HBM traffic of one forward from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command,
beside the algorithmic bytes of each layer (input + weights + fused residual read; output):

    python tools/pmc_layers.py <fetch_results.db> <write_results.db> [batch] [size] [f32|bf16] > profiles/rNN_layer_traffic_*.txt

Takes the last whole detector forward of each pass (the MFMA-conv launches behind the last image-stem launch; run the
command with ``--no-accuracy --no-bf16-line --no-batch-sweep --no-cpu-baseline`` so that every forward is the batch asked for); a slab-reduce launch (split-K /
tail split) is charged to the conv launch in front of it.  Units / corrections as in tools/pmc_traffic.py (KiB; FETCH_SIZE
doubled).  The 1x1 64->32 layer on the 208x208 map is the calibration point: its input (354 MB at batch 32, fp32) cannot be
resident in the 256 MB Infinity Cache, nothing is read twice, so fetch = in + weights if the correction is right."""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

MAIN = ("conv_igemm", "conv3x3_p8", "conv1x1_ws", "conv3x3_ws", "conv_kw")
SECOND = ("conv_tail_reduce", "conv_splitk_reduce")


def layers(batch, size, elt):
    """(name, in_bytes, w_bytes, res_bytes, out_bytes, flops) of every MFMA conv of yolov3.cfg, in launch order."""
    from millieye_amd import cfgs
    blocks = []
    for line in cfgs.yolov3_cfg_text().splitlines():
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("["):
            blocks.append({"type": line[1:-1]})
        else:
            k, v = line.split("=", 1)
            blocks[-1][k.strip()] = v.strip()
    blocks = blocks[1:]
    out, outs, c, h = [], [], 3, size
    for i, b in enumerate(blocks):
        t = b["type"]
        if t == "convolutional":
            k, s, co = int(b["size"]), int(b["stride"]), int(b["filters"])
            ho = h // s
            if c > 4:
                out.append([f"conv{i} {k}x{k}/{s} {c}->{co} @{h}", batch * h * h * c * elt, co * k * k * c * elt, 0,
                            batch * ho * ho * co * elt, 2 * batch * ho * ho * co * k * k * c])
            c, h = co, ho
        elif t == "shortcut":
            out[-1][3] = batch * h * h * c * elt
        elif t == "route":
            ls = [int(x) for x in b["layers"].split(",")]
            ls = [l if l >= 0 else i + l for l in ls]
            c, h = sum(outs[l][0] for l in ls), outs[ls[0]][1]
        elif t == "upsample":
            out[-1][4] *= 4  # fused into the epilogue of the conv in front of it
            h *= 2
        outs.append((c, h))
    return out


def launches(db, counter, n):
    """The MFMA-conv launches of the last whole detector forward: the last stem launch (cin 3: ``conv_stem*``) that is
    followed by at least n MFMA-conv launches before the next stem launch (the radar CNN of the heads has a stem too)."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, value, start, end from counters_collection where counter_name = ? order by start",
                       (counter,)).fetchall()
    runs = [[]]
    for name, value, start, end in rows:
        if "conv_stem" in name or "conv_smallcin" in name:
            runs.append([])
        elif any(m in name for m in MAIN):
            runs[-1].append([name, value, end - start])
        elif any(m in name for m in SECOND) and runs[-1]:
            runs[-1][-1][1] += value
            runs[-1][-1][2] += end - start
    whole = [r for r in runs[1:] if len(r) >= n]
    if not whole:
        raise SystemExit(f"no forward with {n} MFMA-conv launches behind a stem launch in {db}")
    return whole[-1][:n]


def main():
    fetch_db, write_db = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    size = int(sys.argv[4]) if len(sys.argv) > 4 else 416
    elt = 2 if (len(sys.argv) > 5 and sys.argv[5] != "f32") else 4
    ls = layers(batch, size, elt)
    f = launches(fetch_db, "FETCH_SIZE", len(ls))
    w = launches(write_db, "WRITE_SIZE", len(ls))
    print(f"# per-layer HBM traffic of the last forward (MB; fetch = FETCH_SIZE KiB x 1024 x 2, write = WRITE_SIZE KiB x 1024), "
          f"batch {batch}, {size}x{size}, {elt} B / element")
    print(f"{'layer':34s} {'in+w+res':>9s} {'fetch':>8s} {'ratio':>6s} {'out':>8s} {'write':>8s} {'ratio':>6s} {'us (pmc pass)':>13s}  kernel")
    ta = tf = to = tw = 0.0
    for (name, i_b, w_b, r_b, o_b, _fl), (kn, fv, dur), (_kn2, wv, _d2) in zip(ls, f, w):
        rd, fb, wb = (i_b + w_b + r_b) / 1e6, fv * 1024 * 2 / 1e6, wv * 1024 / 1e6
        ta, tf, to, tw = ta + rd, tf + fb, to + o_b / 1e6, tw + wb
        kshort = kn.split("<")[0].split("::")[-1] + "<" + kn.split("<", 1)[1].split(">")[0] + ">" if "<" in kn else kn
        print(f"{name:34s} {rd:9.1f} {fb:8.1f} {fb / rd:6.2f} {o_b / 1e6:8.1f} {wb:8.1f} {wb / (o_b / 1e6):6.2f} {dur / 1e3:13.1f}  {kshort[:60]}")
    n = len(ls)
    print(f"{'sum':34s} {ta:9.1f} {tf:8.1f} {tf / ta:6.2f} {to:8.1f} {tw:8.1f} {tw / to:6.2f}")
    print(f"per launch: algorithmic {(ta + to) / n:.1f} MB, measured {(tf + tw) / n:.1f} MB = {(tf + tw) / (ta + to):.2f}x")


if __name__ == "__main__":
    main()
