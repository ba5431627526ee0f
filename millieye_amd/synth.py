"""Deterministic, library-version-independent synthetic tensors.

There are no weights or datasets for this path (reference ``weights/`` and
``data/`` are empty placeholders), so tests, golden fixtures and ``bench.py`` use
synthetic frames / radar maps / weights.  Fixtures only store *outputs*; the
inputs are regenerated here, therefore the generator must give bit-identical
float32 values on every machine: it is a counter-based integer hash
(splitmix64) followed by exact integer -> float conversions only (no libm).

``normal`` is an Irwin-Hall(12) approximation (sum of twelve 16-bit uniforms):
bounded to +-6 sigma, exact in float64, which is all a weight init needs.
"""
import numpy as np

__all__ = ["tag_seed", "uniform", "normal", "fill_state_dict", "fill_darknet_", "trained_like_"]

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def tag_seed(tag):
    """FNV-1a 64-bit hash of a string tag -> python int seed."""
    h = 0xCBF29CE484222325
    for b in str(tag).encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _bits(tag, count, stream):
    seed = np.uint64(tag_seed(tag))
    idx = np.arange(count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        ctr = (idx * np.uint64(0xD1342543DE82EF95) + seed + np.uint64(stream) * np.uint64(0xA24BAED4963EE407)) & _M64
    return _splitmix64(ctr)


def uniform(tag, shape, lo=0.0, hi=1.0):
    """float32 array, U[lo, hi) from 24 hashed bits per element."""
    count = int(np.prod(shape)) if len(shape) else 1
    u24 = (_bits(tag, count, 0) >> np.uint64(40)).astype(np.float64)  # exact
    u = u24 / 16777216.0
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(tag, shape, mean=0.0, std=1.0):
    """float32 array, approximately N(mean, std^2) (Irwin-Hall of 12 x 16-bit uniforms)."""
    count = int(np.prod(shape)) if len(shape) else 1
    total = np.zeros(count, dtype=np.int64)
    for stream in range(3):  # 3 hashes x 4 sixteen-bit lanes = 12 uniforms
        b = _bits(tag, count, stream + 1)
        for lane in range(4):
            total += ((b >> np.uint64(16 * lane)) & np.uint64(0xFFFF)).astype(np.int64)
    z = total.astype(np.float64) / 65536.0 - 6.0  # mean 6, var 12 * (1/12) = 1
    return (mean + std * z).astype(np.float32).reshape(shape)


def fill_state_dict(module, tag, conv_std=None, linear_std=None, bn_weight=(0.8, 1.2), bn_bias_std=0.1,
                    bn_mean_std=0.1, bn_var=(0.5, 1.5), bias_std=0.1):
    """Overwrite every parameter / buffer of ``module`` (a torch ``nn.Module``) in place
    with deterministic values derived from ``tag`` and the state-dict key.

    Distribution is chosen per key suffix so activations stay O(1) through deep
    stacks: conv/linear weights N(0, 2/fan_in) (He) unless ``conv_std`` /
    ``linear_std`` override, BN affine near identity, running stats near (0, 1).
    """
    import torch

    sd = module.state_dict()
    with torch.no_grad():
        for key, t in sd.items():
            shape = tuple(t.shape)
            k = f"{tag}/{key}"
            if key.endswith("num_batches_tracked"):
                t.zero_()
                continue
            if key.endswith("running_mean"):
                arr = normal(k, shape, 0.0, bn_mean_std)
            elif key.endswith("running_var"):
                arr = uniform(k, shape, bn_var[0], bn_var[1])
            elif key.endswith(".bias"):
                arr = normal(k, shape, 0.0, bn_bias_std if _is_bn(sd, key) else bias_std)
            elif key.endswith(".weight"):
                if _is_bn(sd, key):
                    arr = uniform(k, shape, bn_weight[0], bn_weight[1])
                elif t.dim() == 4:
                    fan_in = shape[1] * shape[2] * shape[3]
                    std = conv_std if conv_std is not None else (2.0 / fan_in) ** 0.5
                    arr = normal(k, shape, 0.0, std)
                elif t.dim() == 2:
                    std = linear_std if linear_std is not None else (2.0 / shape[1]) ** 0.5
                    arr = normal(k, shape, 0.0, std)
                else:
                    arr = normal(k, shape, 0.0, 0.1)
            else:
                arr = normal(k, shape, 0.0, 0.1)
            t.copy_(torch.from_numpy(arr))
    return module


def fill_darknet_(darknet, tag):
    """Deterministic Darknet weights whose activations stay O(1) through all 75 conv layers.

    A plain He init explodes through Darknet-53 (23 residual adds, BN in eval mode with unit
    running variance does not normalise anything): outputs reach 1e4 and every sigmoid of the
    YOLO heads saturates, which would make parity tests blind.  A trained checkpoint does not do
    that because each BatchNorm's running variance matches the variance of its input.  This
    routine emulates that without running the network: it propagates the second moment ``q`` of
    the activations through the cfg graph analytically (conv: var = fan_in * std_w^2 * q; leaky:
    x0.505; shortcut: sum; route: channel-weighted mean) and sets every BN's ``running_var`` to
    the predicted variance of its conv output (x U(0.8, 1.25)), so each block re-normalises.
    """
    import torch

    defs = darknet.module_defs
    q = [None] * len(defs)  # second moment of each module's output
    q_in = 1.0 / 3.0        # frames are U[0,1)
    chans = [int(darknet.hyperparams["channels"])]
    with torch.no_grad():
        for i, d in enumerate(defs):
            kind = d["type"]
            prev = q[i - 1] if i > 0 else q_in
            if kind == "convolutional":
                seq = darknet.module_list[i]
                conv = seq[0]
                cout, cin, kh, kw = conv.weight.shape
                fan_in = cin * kh * kw
                std = (2.0 / fan_in) ** 0.5
                k = f"{tag}/module_list.{i}"
                conv.weight.copy_(torch.from_numpy(normal(k + ".w", (cout, cin, kh, kw), 0.0, std)))
                var_c = fan_in * std * std * prev
                if int(d["batch_normalize"]):
                    bn = seq[1]
                    bn.weight.copy_(torch.from_numpy(uniform(k + ".g", (cout,), 0.8, 1.2)))
                    bn.bias.copy_(torch.from_numpy(normal(k + ".b", (cout,), 0.0, 0.1)))
                    bn.running_mean.copy_(torch.from_numpy(normal(k + ".m", (cout,), 0.0, 0.1 * var_c ** 0.5)))
                    bn.running_var.copy_(torch.from_numpy(uniform(k + ".v", (cout,), 0.8 * var_c, 1.25 * var_c)))
                    bn.num_batches_tracked.zero_()
                    out_q = 1.08  # E[g^2] * E[var_c / running_var] + bias^2
                else:
                    conv.bias.copy_(torch.from_numpy(normal(k + ".cb", (cout,), 0.0, 0.1)))
                    out_q = var_c + 0.01
                if d["activation"] == "leaky":
                    out_q *= 0.505
                q[i] = out_q
                chans.append(cout)
            elif kind == "maxpool":
                q[i] = prev * 1.6  # max of four positively correlated samples
                chans.append(chans[-1])
            elif kind == "upsample":
                q[i] = prev
                chans.append(chans[-1])
            elif kind == "route":
                idx = [int(v) for v in d["layers"].split(",")]
                idx = [i + v if v < 0 else v for v in idx]
                cs = [chans[1:][j] for j in idx]
                q[i] = sum(q[j] * c for j, c in zip(idx, cs)) / sum(cs)
                chans.append(sum(cs))
            elif kind == "shortcut":
                frm = int(d["from"])
                j = i + frm if frm < 0 else frm
                q[i] = q[i - 1] + q[j]
                chans.append(chans[-1])
            elif kind == "yolo":
                q[i] = prev
                chans.append(chans[-1])
    return darknet


def fill_network_(net, tag, trained_like=True, **trained_kwargs):
    """Deterministic weights for a whole fusion ``Network`` (product or reference instance - the
    module / parameter names are identical): calibrated detector (+ trained-like detection
    convs), He-style heads with non-trivial BatchNorm statistics."""
    fill_darknet_(net.base_detector, tag + "/det")
    if trained_like:
        trained_like_(net.base_detector, tag + "/det/trained", **trained_kwargs)
    for name in ("img_cnn_layers", "radar_cnn_layers", "fcn_layers", "refinement_head", "ensemble_head"):
        if hasattr(net, name):  # stage 3 has img_/radar_cnn_layers, stage 2 (module2_mixed) fcn_layers
            fill_state_dict(getattr(net, name), f"{tag}/{name}")
    return net


def radar_inputs(tag, n, side, boxes_per_image=2, density=0.02):
    """Synthetic radar heat maps ``[n,3,side,side]`` (about ``density`` non-zeros in [0,1], the
    measured sparsity of real maps) and radar box proposals ``[n*b,5]`` =
    (image_i, x1, y1, x2, y2) in [0,1] units with x1 < x2, y1 < y2."""
    m = uniform(tag + "/maps", (n, 3, side, side))
    mask = uniform(tag + "/mask", (n, 3, side, side)) < density
    maps = np.where(mask, m, 0.0).astype(np.float32)
    c = uniform(tag + "/bc", (n * boxes_per_image, 2), 0.2, 0.8)
    half = uniform(tag + "/bs", (n * boxes_per_image, 2), 0.05, 0.2)
    idx = np.repeat(np.arange(n, dtype=np.float32), boxes_per_image)[:, None]
    boxes = np.concatenate([idx, c - half, c + half], 1).astype(np.float32)
    return maps, boxes


def _is_bn(sd, key):
    stem = key.rsplit(".", 1)[0]
    return (stem + ".running_var") in sd


_DET_GAIN = {107: [4.0, 1.93, 1.34], 24: [0.30, 0.36]}  # yolov3.cfg / yolov3-tiny cfgs


def trained_like_(darknet, tag="trained", obj_bias=-4.0, obj_std=2.0, wh_std=0.5, cls_bias=-2.0, gains=None,
                  cls0_bias=1.0):
    """Give the detection convolutions of a ``Darknet`` (in place) the output statistics of
    a *trained* detector: objectness logits around
    ``obj_bias`` so that only a few percent of rows pass ``conf_thresh``, ``tw/th`` ~
    N(0, wh_std) so boxes stay box-sized, class-0 logit boosted so class-0 proposals exist.

    The detection conv is the ``[convolutional]`` block right before every ``[yolo]``;
    its weights are rescaled so that (for O(1) inputs) the logit spread is ~1, then
    per-channel biases set the means.
    """
    import torch

    defs = darknet.module_defs
    if gains is None:
        # measured std of unit-scale detection logits on fill_darknet_ weights at 416x416
        # (the backbone's activations drift upward a little): per [yolo] block, cfg order
        gains = _DET_GAIN.get(len(defs), [1.0] * 8)
    yolo_no = 0
    for i, d in enumerate(defs):
        if d["type"] != "yolo":
            continue
        conv = darknet.module_list[i - 1][0]
        num_classes = int(d["classes"])
        per = num_classes + 5
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        gain = gains[yolo_no]
        yolo_no += 1
        with torch.no_grad():
            w = normal(f"{tag}/det{i}/w", tuple(conv.weight.shape), 0.0, 1.0 / (cin ** 0.5) / gain)
            b = np.zeros(cout, dtype=np.float32)
            scale = np.ones(cout, dtype=np.float32)
            for a in range(cout // per):
                base = a * per
                scale[base + 0: base + 2] = 1.0
                scale[base + 2: base + 4] = wh_std
                scale[base + 4] = obj_std
                b[base + 4] = obj_bias
                scale[base + 5: base + per] = 1.0
                b[base + 5: base + per] = cls_bias
                b[base + 5] = cls0_bias
            w = w * scale[:, None, None, None]
            conv.weight.copy_(torch.from_numpy(w.astype(np.float32)))
            conv.bias.copy_(torch.from_numpy(b))
    return darknet
