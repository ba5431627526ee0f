"""ORACLE (test infrastructure - never imported by the product path).

CPU restatement, in stock PyTorch fp32 ops, of the reference detector forward:

  * cfg parsing            module3_our_dataset/utils/parse_config.py:3-21
  * conv/BN/leaky blocks   module3_our_dataset/yolov3/models.py:22-41
  * maxpool (+ zero pad)   models.py:43-49        * upsample  models.py:82-92
  * route / shortcut       models.py:256-260      * YOLO decode models.py:132-179
  * Darknet.forward        models.py:247-267 (returns featuremap, yolo_outputs)

It is a *functional* evaluator: ``darknet_forward(cfg_text, state_dict, x)`` walks the parsed
blocks and applies ``torch.nn.functional`` ops to tensors taken from a state dict with the
reference's key names - the same library calls (MKLDNN conv, native batch_norm ...) the
reference's CPU path ends up in, so it doubles as the timed CPU baseline in bench.py
(``cpu_baseline.kind = "port"``).

Pinning: tests/golden/darknet_*.npz were produced by importing the real reference
(tests/golden/make_golden.py, Appendix-E recipe) on the same deterministic weights/inputs;
tests/test_oracle_golden.py checks this file against them.
"""
import torch
import torch.nn.functional as F


def parse_cfg_text(text):
    """cfg text -> list of dict blocks ([net] first).  Values stay strings; convolutional blocks
    default ``batch_normalize`` to integer 0 (reference parse_config.py:14-15)."""
    blocks = []
    for raw in text.split("\n"):
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("["):
            blocks.append({"type": line[1:-1].rstrip()})
            if blocks[-1]["type"] == "convolutional":
                blocks[-1]["batch_normalize"] = 0
        else:
            k, v = line.split("=")
            blocks[-1][k.rstrip()] = v.strip()
    return blocks


def yolo_decode(x, anchors, num_classes, img_dim):
    """Reference YOLOLayer inference branch (models.py:132-179) on an NCHW map."""
    n, g = x.size(0), x.size(2)
    na = len(anchors)
    pred = x.view(n, na, num_classes + 5, g, g).permute(0, 1, 3, 4, 2).contiguous()
    sx = torch.sigmoid(pred[..., 0])
    sy = torch.sigmoid(pred[..., 1])
    tw, th = pred[..., 2], pred[..., 3]
    conf = torch.sigmoid(pred[..., 4])
    cls = torch.sigmoid(pred[..., 5:])
    stride = img_dim / g
    gx = torch.arange(g).repeat(g, 1).view(1, 1, g, g).float()
    gy = torch.arange(g).repeat(g, 1).t().view(1, 1, g, g).float()
    scaled = torch.tensor([(aw / stride, ah / stride) for aw, ah in anchors], dtype=torch.float32)
    aw = scaled[:, 0:1].view(1, na, 1, 1)
    ah = scaled[:, 1:2].view(1, na, 1, 1)
    boxes = torch.empty(pred[..., :4].shape, dtype=torch.float32)
    boxes[..., 0] = sx + gx
    boxes[..., 1] = sy + gy
    boxes[..., 2] = torch.exp(tw) * aw
    boxes[..., 3] = torch.exp(th) * ah
    return torch.cat((boxes.view(n, -1, 4) * stride, conf.view(n, -1, 1), cls.view(n, -1, num_classes)), -1)


def _anchors_of(block):
    flat = [int(v) for v in block["anchors"].split(",")]
    pairs = [(flat[i], flat[i + 1]) for i in range(0, len(flat), 2)]
    return [pairs[int(m)] for m in block["mask"].split(",")]


def _readers(blocks):
    """module index -> indices of the modules that read its output"""
    readers = [[] for _ in blocks]
    for i, b in enumerate(blocks):
        if b["type"] == "route":
            srcs = [int(l) for l in b["layers"].split(",")]
        elif b["type"] == "shortcut":
            srcs = [-1, int(b["from"])]
        else:
            srcs = [-1]
        for s in srcs:
            j = i + s if s < 0 else s
            if j >= 0:
                readers[j].append(i)
    return readers


def darknet_forward(cfg_text, state_dict, x, tap_module=8, prefix="module_list.", return_layers=False, storage="f32",
                    training=False):
    """Evaluate the detector.  ``state_dict`` keys: ``{prefix}{i}.conv_{i}.weight`` etc.
    Returns ``(featuremap or None, yolo_outputs [N,R,5+C])`` (+ per-module outputs on request).

    ``storage="bf16"`` / ``"f16"`` restates the build's 16-bit storage modes (BASELINE configs[2]/[4]; no reference counterpart - the
    reference is fp32 only, so this mode is pinned to nothing but the fp32 path it approximates): same fp32 library ops,
    with one round-to-nearest-even to bfloat16 at every point where millieye_amd/csrc/conv_bf16.hip stores 16-bit data -
    the frame and the weights of every convolution, and every activation written to memory, i.e. after
    conv+BN+LeakyReLU (and after the [shortcut] add when that convolution feeds only the shortcut - the add is then part of
    the same kernel and sees the unrounded value).  Detection convolutions (read only by a [yolo] block) stay fp32.

    ``training=True``: BatchNorm on batch statistics, as ``Darknet.forward(x)`` computes under ``model.train()`` in the
    reference (yolov3/models.py:35,247-267); the running statistics of ``state_dict`` are updated IN PLACE with momentum 0.9
    like the module's buffers (pass a copy to keep the originals)."""
    blocks = parse_cfg_text(cfg_text)[1:]
    img_dim = x.shape[2]
    outs, yolo = [], []
    feat = None
    bf16 = storage in ("bf16", "f16")  # any 16-bit storage mode: same rounding points, other format
    if storage not in ("f32", "bf16", "f16"):
        raise ValueError(storage)
    half = torch.float16 if storage == "f16" else torch.bfloat16
    readers = _readers(blocks)

    def q(t):
        return t.to(half).to(torch.float32) if bf16 else t

    pending = None  # unrounded output of a conv fused with the following shortcut
    with torch.no_grad():
        x = q(x)  # the stem reads the frame as 16-bit MFMA operands
        for i, b in enumerate(blocks):
            kind = b["type"]
            if kind == "convolutional":
                k = int(b["size"])
                w = state_dict[f"{prefix}{i}.conv_{i}.weight"]
                bias = state_dict.get(f"{prefix}{i}.conv_{i}.bias")
                if bf16:
                    w = q(w)
                x = F.conv2d(x, w, bias, stride=int(b["stride"]), padding=(k - 1) // 2)
                if int(b["batch_normalize"]):
                    p = f"{prefix}{i}.batch_norm_{i}."
                    x = F.batch_norm(x, state_dict[p + "running_mean"], state_dict[p + "running_var"],
                                     state_dict[p + "weight"], state_dict[p + "bias"], training, 0.9, 1e-5)
                if b["activation"] == "leaky":
                    x = F.leaky_relu(x, 0.1)
                if bf16:
                    nxt = blocks[i + 1] if i + 1 < len(blocks) else None
                    fused = (nxt is not None and nxt["type"] == "shortcut" and readers[i] == [i + 1] and i != tap_module
                             and w.shape[1] > 4 and i + 1 + int(nxt["from"]) != i)
                    if fused:
                        pending = x
                    if not all(blocks[r]["type"] == "yolo" for r in readers[i]) or not readers[i]:
                        x = q(x)
                if i == tap_module:
                    feat = x
            elif kind == "maxpool":
                k, s = int(b["size"]), int(b["stride"])
                if k == 2 and s == 1:
                    x = F.pad(x, (0, 1, 0, 1), value=0.0)
                x = F.max_pool2d(x, k, s, (k - 1) // 2)
            elif kind == "upsample":
                x = F.interpolate(x, scale_factor=int(b["stride"]), mode="nearest")
            elif kind == "route":
                x = torch.cat([outs[int(l)] for l in b["layers"].split(",")], 1)
            elif kind == "shortcut":
                x = q((outs[-1] if pending is None else pending) + outs[int(b["from"])])
                pending = None
            elif kind == "yolo":
                x = yolo_decode(x, _anchors_of(b), int(b["classes"]), img_dim)
                yolo.append(x)
            else:
                raise ValueError(kind)
            outs.append(x)
    res = (feat, torch.cat(yolo, 1))
    return res + (outs,) if return_layers else res


# ---------------------------------------------------------------------------------------------------
# detector training step: Darknet.forward(x, targets) -> loss.backward() (models.py:181-267 under autograd, BatchNorm
# in eval mode as every reference script keeps it).  Pinned by tests/golden/yololoss_*.npz (loss + every gradient).
# ---------------------------------------------------------------------------------------------------
def _wh_iou(wh1, wh2):
    """bbox_wh_iou (utils/utils.py:239-245)."""
    wh2 = wh2.t()
    inter = torch.min(wh1[0], wh2[0]) * torch.min(wh1[1], wh2[1])
    return inter / ((wh1[0] * wh1[1] + 1e-16) + wh2[0] * wh2[1] - inter)


def yolo_loss(raw, anchors, num_classes, img_dim, targets, ignore_thres=0.5, obj_scale=1, noobj_scale=100):
    """Loss of one YOLOLayer (models.py:132-214) from its raw input map ``raw`` [N, A*(5+C), G, G] (NCHW)."""
    n, g, na = raw.shape[0], raw.shape[2], len(anchors)
    pred = raw.view(n, na, num_classes + 5, g, g).permute(0, 1, 3, 4, 2).contiguous()
    x, y = torch.sigmoid(pred[..., 0]), torch.sigmoid(pred[..., 1])
    w, h = pred[..., 2], pred[..., 3]
    conf, cls = torch.sigmoid(pred[..., 4]), torch.sigmoid(pred[..., 5:])
    stride = img_dim / g
    sa = torch.tensor([(aw / stride, ah / stride) for aw, ah in anchors], dtype=torch.float32)
    # build_targets (utils/utils.py:381-440): index bookkeeping, no gradient
    with torch.no_grad():
        obj = torch.zeros(n, na, g, g, dtype=torch.bool)
        noobj = torch.ones(n, na, g, g, dtype=torch.bool)
        tx, ty, tw, th = (torch.zeros(n, na, g, g) for _ in range(4))
        tcls = torch.zeros(n, na, g, g, num_classes)
        tb = targets[:, 2:6] * g
        gxy, gwh = tb[:, :2], tb[:, 2:]
        ious = torch.stack([_wh_iou(a, gwh) for a in sa])
        best_n = ious.max(0)[1]
        b, labels = targets[:, :2].long().t()
        gi, gj = gxy.long().t()
        obj[b, best_n, gj, gi] = True
        noobj[b, best_n, gj, gi] = False
        for k, a_ious in enumerate(ious.t()):
            noobj[b[k], a_ious > ignore_thres, gj[k], gi[k]] = False
        tx[b, best_n, gj, gi] = gxy[:, 0] - gxy[:, 0].floor()
        ty[b, best_n, gj, gi] = gxy[:, 1] - gxy[:, 1].floor()
        tw[b, best_n, gj, gi] = torch.log(gwh[:, 0] / sa[best_n][:, 0] + 1e-16)
        th[b, best_n, gj, gi] = torch.log(gwh[:, 1] / sa[best_n][:, 1] + 1e-16)
        tcls[b, best_n, gj, gi, labels] = 1
        tconf = obj.float()
    mse, bce = F.mse_loss, F.binary_cross_entropy
    loss = (mse(x[obj], tx[obj]) + mse(y[obj], ty[obj]) + mse(w[obj], tw[obj]) + mse(h[obj], th[obj])
            + obj_scale * bce(conf[obj], tconf[obj]) + noobj_scale * bce(conf[noobj], tconf[noobj])
            + bce(cls[obj], tcls[obj]))
    return loss


def yolo_loss_terms(raw_nhwc, anchors, num_classes, img_dim, targets, ignore_thres=0.5, obj_scale=1, noobj_scale=100):
    """Everything ``YOLOLayer.forward`` computes besides the decoded rows when it is given targets (models.py:181-232 with
    build_targets, utils/utils.py:381-440), from the raw map ``raw_nhwc`` [N, G, G, A*(5+C)]: ``(total_loss, metrics,
    dense)`` - the reference's metrics dict and the dense build_targets tensors (obj / noobj masks, tx, ty, tw, th, tcls,
    tconf, class_mask, iou_scores).  The reference's own statements, one torch op each (its per-target python loop included);
    the checker of ``me_yolo_loss_fwd_f32``.  Loss and gradients of the same code are pinned by ``yolo_loss`` above against
    tests/golden/yololoss_*.npz, the metrics by the GPU test against the same fixtures."""
    n, g, na = raw_nhwc.shape[0], raw_nhwc.shape[1], len(anchors)
    with torch.no_grad():
        pred = raw_nhwc.reshape(n, g, g, na, num_classes + 5).permute(0, 3, 1, 2, 4)
        x, y = torch.sigmoid(pred[..., 0]), torch.sigmoid(pred[..., 1])
        w, h = pred[..., 2], pred[..., 3]
        pred_conf, pred_cls = torch.sigmoid(pred[..., 4]), torch.sigmoid(pred[..., 5:])
        stride = img_dim / g
        grid_x = torch.arange(g).repeat(g, 1).view(1, 1, g, g).float()
        grid_y = torch.arange(g).repeat(g, 1).t().view(1, 1, g, g).float()
        sa = torch.tensor([(aw / stride, ah / stride) for aw, ah in anchors], dtype=torch.float32)
        pred_boxes = torch.stack((x + grid_x, y + grid_y, torch.exp(w) * sa[:, 0].view(1, na, 1, 1),
                                  torch.exp(h) * sa[:, 1].view(1, na, 1, 1)), -1)
        cells = (n, na, g, g)
        obj = torch.zeros(cells, dtype=torch.bool)
        noobj = torch.ones(cells, dtype=torch.bool)
        class_mask, iou_scores, tx, ty, tw, th = (torch.zeros(cells) for _ in range(6))
        tcls = torch.zeros(cells + (num_classes,))
        if targets.shape[0]:
            tb = targets[:, 2:6] * g
            gxy, gwh = tb[:, :2], tb[:, 2:]
            ious = torch.stack([_wh_iou(a, gwh) for a in sa])
            best_n = ious.max(0)[1]
            b, labels = targets[:, :2].long().t()
            gi, gj = gxy.long().t()
            obj[b, best_n, gj, gi] = True
            noobj[b, best_n, gj, gi] = False
            for k, a_ious in enumerate(ious.t()):
                noobj[b[k], a_ious > ignore_thres, gj[k], gi[k]] = False
            tx[b, best_n, gj, gi] = gxy[:, 0] - gxy[:, 0].floor()
            ty[b, best_n, gj, gi] = gxy[:, 1] - gxy[:, 1].floor()
            tw[b, best_n, gj, gi] = torch.log(gwh[:, 0] / sa[best_n][:, 0] + 1e-16)
            th[b, best_n, gj, gi] = torch.log(gwh[:, 1] / sa[best_n][:, 1] + 1e-16)
            tcls[b, best_n, gj, gi, labels] = 1
            class_mask[b, best_n, gj, gi] = (pred_cls[b, best_n, gj, gi].argmax(-1) == labels).float()
            pb = pred_boxes[b, best_n, gj, gi]                                 # bbox_iou(..., x1y1x2y2=False), utils.py:173-200
            b1x1, b1x2 = pb[:, 0] - pb[:, 2] / 2, pb[:, 0] + pb[:, 2] / 2
            b1y1, b1y2 = pb[:, 1] - pb[:, 3] / 2, pb[:, 1] + pb[:, 3] / 2
            b2x1, b2x2 = tb[:, 0] - tb[:, 2] / 2, tb[:, 0] + tb[:, 2] / 2
            b2y1, b2y2 = tb[:, 1] - tb[:, 3] / 2, tb[:, 1] + tb[:, 3] / 2
            inter = torch.clamp(torch.min(b1x2, b2x2) - torch.max(b1x1, b2x1) + 1, min=0) * \
                torch.clamp(torch.min(b1y2, b2y2) - torch.max(b1y1, b2y1) + 1, min=0)
            a1, a2 = (b1x2 - b1x1 + 1) * (b1y2 - b1y1 + 1), (b2x2 - b2x1 + 1) * (b2y2 - b2y1 + 1)
            iou_scores[b, best_n, gj, gi] = inter / (a1 + a2 - inter + 1e-16)
        tconf = obj.float()
        mse, bce = F.mse_loss, F.binary_cross_entropy
        loss_x, loss_y = mse(x[obj], tx[obj]), mse(y[obj], ty[obj])
        loss_w, loss_h = mse(w[obj], tw[obj]), mse(h[obj], th[obj])
        loss_conf = obj_scale * bce(pred_conf[obj], tconf[obj]) + noobj_scale * bce(pred_conf[noobj], tconf[noobj])
        loss_cls = bce(pred_cls[obj], tcls[obj])
        total = loss_x + loss_y + loss_w + loss_h + loss_conf + loss_cls
        conf50, iou50, iou75 = (pred_conf > 0.5).float(), (iou_scores > 0.5).float(), (iou_scores > 0.75).float()
        detected = conf50 * class_mask * tconf
        metrics = {"loss": total.item(), "x": loss_x.item(), "y": loss_y.item(), "w": loss_w.item(), "h": loss_h.item(),
                   "conf": loss_conf.item(), "cls": loss_cls.item(), "cls_acc": (100 * class_mask[obj].mean()).item(),
                   "recall50": (torch.sum(iou50 * detected) / (obj.sum() + 1e-16)).item(),
                   "recall75": (torch.sum(iou75 * detected) / (obj.sum() + 1e-16)).item(),
                   "precision": (torch.sum(iou50 * detected) / (conf50.sum() + 1e-16)).item(),
                   "conf_obj": pred_conf[obj].mean().item(), "conf_noobj": pred_conf[noobj].mean().item(), "grid_size": g}
        dense = dict(obj=obj, noobj=noobj, tx=tx, ty=ty, tw=tw, th=th, tcls=tcls, tconf=tconf, class_mask=class_mask,
                     iou_scores=iou_scores, n_obj=int(obj.sum()), n_noobj=int(noobj.sum()))
    return total, metrics, dense


def darknet_train_step(cfg_text, state_dict, x, targets, prefix="module_list.", training=False, storage="f32"):
    """Summed YOLO loss of every scale + its gradient w.r.t. every detector parameter.  ``training`` selects the
    BatchNorm mode (False: running statistics, what every reference script uses; True: batch statistics, momentum 0.9).
    Returns ``(loss, {name: grad})`` and, with ``training``, ``{name: updated running statistic}`` as a third item;
    ``state_dict`` is not modified.

    ``storage="bf16"`` / ``"f16"`` restates the build's 16-bit TRAINING step (millieye_amd/detector_train16.py; no reference
    counterpart - the reference trains in fp32 only, so this mode is pinned to nothing but the fp32 step it approximates): the same
    fp32 autograd graph with one round-to-nearest-even to the storage type at every point where that path stores 16-bit data -
    forward: the frame, the weights every convolution multiplies (the fp32 masters receive the gradients: straight-through),
    every stored activation (after conv + BN + activation; after each [shortcut] add - the training forward keeps every module
    output, so the add is its own launch and reads ROUNDED operands, unlike the inference plans' fused epilogue); backward: the
    gradient w.r.t. every stored 16-bit activation (after all its readers' contributions are added) and the gradient w.r.t.
    every convolution's raw output (``me_affine_act_bwd_h16`` / the converted result of ``me_bn_train_bwd_f32``).  Detection
    convolutions (read only by a [yolo] block) and the YOLO loss stay fp32, as do all parameter gradients and BatchNorm statistics."""
    blocks = parse_cfg_text(cfg_text)[1:]
    img_dim = x.shape[2]
    if storage not in ("f32", "bf16", "f16"):
        raise ValueError(storage)
    low = storage != "f32"
    half = torch.float16 if storage == "f16" else torch.bfloat16
    readers = _readers(blocks)

    def q(t):
        return t.to(half).to(torch.float32)

    def stored(t):
        """value rounded to the storage type (straight-through), gradient w.r.t. the stored tensor rounded once"""
        if not low:
            return t
        t = t + (q(t) - t).detach()
        if t.requires_grad:
            t.register_hook(q)
        return t

    P = {k: v.detach().clone().requires_grad_(True) for k, v in state_dict.items()
         if v.dtype == torch.float32 and "running_" not in k}
    B = {k: v.detach().clone() for k, v in state_dict.items() if "running_" in k}
    outs = []
    loss = 0
    if low:
        x = q(x)
    for i, b in enumerate(blocks):
        kind = b["type"]
        if kind == "convolutional":
            k = int(b["size"])
            w = P[f"{prefix}{i}.conv_{i}.weight"]
            detect = bool(readers[i]) and all(blocks[r]["type"] == "yolo" for r in readers[i])
            if low:
                w = w + (q(w) - w).detach()
            x = F.conv2d(x, w, P.get(f"{prefix}{i}.conv_{i}.bias"), stride=int(b["stride"]), padding=(k - 1) // 2)
            if low and not detect:
                x.register_hook(q)   # the activation gradient w.r.t. the convolution's raw output is a stored 16-bit tensor
            if int(b["batch_normalize"]):
                p = f"{prefix}{i}.batch_norm_{i}."
                x = F.batch_norm(x, B[p + "running_mean"], B[p + "running_var"], P[p + "weight"], P[p + "bias"], training,
                                 0.9, 1e-5)
            if b["activation"] == "leaky":
                x = F.leaky_relu(x, 0.1)
            if not detect:
                x = stored(x)
        elif kind == "maxpool":
            k, s = int(b["size"]), int(b["stride"])
            if k == 2 and s == 1:
                x = F.pad(x, (0, 1, 0, 1), value=0.0)
            x = F.max_pool2d(x, k, s, (k - 1) // 2)
        elif kind == "upsample":
            x = F.interpolate(x, scale_factor=int(b["stride"]), mode="nearest")
        elif kind == "route":
            x = torch.cat([outs[int(l)] for l in b["layers"].split(",")], 1)
        elif kind == "shortcut":
            x = stored(outs[-1] + outs[int(b["from"])])
        elif kind == "yolo":
            loss = loss + yolo_loss(x, _anchors_of(b), int(b["classes"]), img_dim, targets)
        outs.append(x)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
    return (loss.detach(), grads, B) if training else (loss.detach(), grads)
