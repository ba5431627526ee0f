"""Stage-2 training step: ``Network.forward(images, targets)`` -> ``(output, loss, metric)`` on the HIP library.

Reference: ``module2_mixed/my_models.py:366-459`` on top of the train-mode forward of ``fcn_layers`` /
``refinement_head`` / ``ensemble_head``, differentiated by torch autograd (``module2_mixed/train.py:138-144``).  As in
stage 3 (``millieye_amd/train_path.py``) the graph is fixed, so forward and backward are explicit launch sequences and
autograd sees one :class:`torch.autograd.Function` over the head parameters:

forward   frozen detector -> NMS -> all-class proposals (as in inference); ``me_conv2d_f32`` + ``me_bn_train_fwd_f32``
          (fcn_layers, batch statistics); ``me_ps_roi_align_f32``; ``me_linear_f32`` for net0 (+ LeakyReLU), the Dropout mask
          (``Network.dropout_generator``: ``me_dropout_mask_u8`` on the device by default; "cpu" draws it with torch's CPU generator
          exactly like aten's CPU dropout - ``empty_like(t).bernoulli_(0.5)`` - for the comparisons with the reference's CPU run)
          through ``me_mask_scale_f32``; ``me_linear_f32`` for net1 / net2 (sigmoid) / fc1 / fc2 (LeakyReLU); host-side IoU labels and
          python-``random`` negative sampling (host-side in the reference too); ``me_m2_loss_f32`` = focal + confidence +
          category + SmoothL1 terms and the gradient seeds.
backward  ``me_act_bwd_f32`` + ``me_gemm_f32`` / MFMA weight gradient / ``me_colsum_f32`` per Linear, the Dropout mask again,
          ``me_ps_roi_align_bwd_f32``, ``me_bn_train_bwd_f32``, 1x1-conv weight gradient.
"""
import random

import numpy as np
import torch

from .. import hip
from ..my_models import _DETECTIONS_PER_IMG, _NMS_THRESH
from ..train_path import (_bn_bwd, _bn_fwd, _colsum, _conv, _f32, _frozen_detector_block, _gemm, _head_named, _issue_prefetch, _ptr,
                          _take_prefetch,
                          iou_labels_vectorized)
from ..utils.utils import xywh2xyxy

LEAKY, SIGMOID, LINEAR = hip.ACT_LEAKY, hip.ACT_SIGMOID, hip.ACT_LINEAR


def _linear(x, ldx, rows, in_f, w, b, act, y, ldy):
    hip.check(hip.lib().me_linear_f32(_ptr(x), ldx, rows, in_f, _ptr(w), _ptr(b), w.shape[0], act, _ptr(y), ldy,
                                      hip.stream_ptr()), "me_linear_f32")
    return y


def _act_bwd(y, dy, rows, c, act):
    dx = torch.empty_like(dy)
    hip.check(hip.lib().me_act_bwd_f32(_ptr(y), c, _ptr(dy), c, _ptr(dx), c, rows, c, act, hip.stream_ptr()), "me_act_bwd_f32")
    return dx


def _linear_bwd(x, rows, in_f, w, dz, need_dx=True, dx_into=None):
    """Gradients of z = x . w^T + b given dz [rows,out]: (dw [out,in], db [out], dx [rows,in] | None)."""
    out_f = w.shape[0]
    dev = x.device
    dw = _f32(dev, out_f, in_f)   # (beta = 0: the products never read C, no zero fill)
    _gemm(1, 0, out_f, in_f, rows, dz, out_f, x, in_f, dw, in_f)
    db = _f32(dev, out_f)
    _colsum(dz, out_f, rows, out_f, db)
    dx = None
    if dx_into is not None:   # dx += dz . w into the gradient another consumer of x has already written
        dx = dx_into
        _gemm(0, 0, rows, in_f, out_f, dz, out_f, w, in_f, dx, in_f, beta=1.0)
    elif need_dx:
        dx = _f32(dev, rows, in_f)
        _gemm(0, 0, rows, in_f, out_f, dz, out_f, w, in_f, dx, in_f)
    return dw, db, dx


_LOSS_W = {}


def _loss_weights(dev, l0, l1):
    key = (str(dev), l0, l1)
    w = _LOSS_W.get(key)
    if w is None:
        w = _LOSS_W[key] = torch.tensor([1.0, 1.0 / l0, 1.0 / l0, 1.0 / l1, 1.0 / l1], dtype=torch.float32).to(dev)
    return w


class _StageTwo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, *params):
        ctx.state = state
        return state["loss"].clone()

    @staticmethod
    def backward(ctx, grad_out):
        grads = _backward(ctx.state, grad_out)
        return (None,) + tuple(grads.get(n) for n in ctx.state["names"])


def forward_train(net, images, targets):
    if not images.is_cuda:
        raise hip.MeError("Network.forward needs CUDA tensors (MI355X); there is no CPU fallback")
    bn = net.fcn_layers.net[1]
    if not (bn.training and net.refinement_head.training):
        raise NotImplementedError("module-2 Network.forward in a mixed mode (fcn_layers and refinement_head must both be in "
                                  "train(), as module2_mixed/train.py:127 sets them, or both in eval())")
    dev, n, size = images.device, images.shape[0], images.shape[-1]
    lib = hip.lib()
    f32 = dict(device=dev, dtype=torch.float32)
    rh, eh = net.refinement_head, net.ensemble_head
    c1 = net.class_num + 1

    with torch.no_grad():
        # frozen detector -> NMS -> all-class proposal rows -> fp32 NHWC copy of the feature tap: the stage-3 path's block
        # (millieye_amd/train_path.py), taken from the look-ahead when the training loop named this batch one call early
        pre = _take_prefetch(net, images)
        if pre is None:
            pre = _frozen_detector_block(net, images)
        nxt = net.__dict__.pop("_next_images", None)
        if nxt is not None:   # (Network.queue_detector_prefetch: the NEXT batch's frozen part beside this batch's host-bound tail)
            _issue_prefetch(net, nxt)
        cols, cap = pre["cols"], pre["cap_img"]
        boxes, n_dev = pre["img_boxes"], pre["n_img_dev"]
        k = int(n_dev.item())
        if k == 0:
            raise hip.MeError("module-2 training step: the detector produced no proposal for this batch")
        boxes = boxes[:k]   # (leading rows of a contiguous buffer: contiguous)
        fh, fw, fc = pre["tap_shape"]
        fm = pre["fm"]
        pix = n * fh * fw
        ws_t = torch.empty(int(lib.me_bn_workspace_bytes(512)) + 256, dtype=torch.uint8, device=dev)
        ws = ws_t.data_ptr() + (-ws_t.data_ptr()) % 256

        # fcn_layers: conv1x1 (+bias) -> BN(train) -> leaky
        icl = net.fcn_layers.net
        w_img = icl[0].weight.detach().reshape(490, fc).contiguous()
        z1 = _f32(dev, pix, 490)
        _conv(fm, fc, n, fh, fw, fc, w_img.view(490, 1, 1, fc), torch.ones(490, **f32), icl[0].bias.detach().contiguous(), 1, 0,
              LINEAR, z1.view(n, fh, fw, 490))
        a1 = _f32(dev, pix, 490)
        st_img = _bn_fwd(z1, 490, pix, 490, icl[1], LEAKY, a1, 490, ws)

        # PS-RoIAlign -> refinement head
        rois = boxes[:, :5].contiguous()
        feat = _f32(dev, k, 490)
        hip.check(lib.me_ps_roi_align_f32(a1.data_ptr(), 490, n, fh, fw, 490, rois.data_ptr(), k, 7, 1.0 / 16, feat.data_ptr(),
                                          hip.stream_ptr()), "me_ps_roi_align_f32")
        w0, b0 = rh.net0[0].weight.detach().contiguous(), rh.net0[0].bias.detach().contiguous()
        w1, b1 = rh.net1[0].weight.detach().contiguous(), rh.net1[0].bias.detach().contiguous()
        w2, b2 = rh.net2[0].weight.detach().contiguous(), rh.net2[0].bias.detach().contiguous()
        e1w, e1b = eh.fc1[0].weight.detach().contiguous(), eh.fc1[0].bias.detach().contiguous()
        e2w, e2b = eh.fc2[0].weight.detach().contiguous(), eh.fc2[0].bias.detach().contiguous()
        t_act = _linear(feat, 490, k, 490, w0, b0, LEAKY, _f32(dev, k, 256), 256)
        # nn.Dropout(0.5), train mode.  Network.dropout_generator names where the keep mask comes from:
        gen = getattr(net, "dropout_generator", "philox")
        if gen == "philox":
            # (default) me_dropout_mask_u8: Philox4x32-10 on the device, keyed by ONE 64-bit draw from torch's CPU generator per step -
            # reproducible under torch.manual_seed, no host work that grows with the proposal count
            seed = int(torch.empty((), dtype=torch.int64).random_())
            mask = torch.empty((k, 256), device=dev, dtype=torch.uint8)
            hip.check(lib.me_dropout_mask_u8(seed, 0.5, k * 256, mask.data_ptr(), hip.stream_ptr()), "me_dropout_mask_u8")
        elif gen == "device":
            # the mask from torch's generator of the GPU - what the reference's own run on a CUDA machine does
            mask = torch.empty((k, 256), device=dev).bernoulli_(0.5).to(torch.uint8)
        elif gen == "cpu":
            # aten's CPU dropout draws empty_like(t).bernoulli_(1 - p) from the default generator: the reference's CPU run (the goldens)
            # bit for bit, at 3.6 ms of host time per step at 1600 proposals - the whole step was bound by it (5.8 ms at batch 8).
            # (the float mask goes to the device as it is and is narrowed THERE: the float -> uint8 conversion of 400 K elements on the
            #  host runs through a 128-thread parallel region on the GPU boxes' CPUs - 34 ms, tools/bernoulli_probe.py)
            mask = torch.empty((k, 256)).bernoulli_(0.5).to(dev).to(torch.uint8)
        else:
            raise ValueError(f"Network.dropout_generator = {gen!r}: expected 'philox', 'device' or 'cpu'")
        hidden = _f32(dev, k, 256)
        hip.check(lib.me_mask_scale_f32(t_act.data_ptr(), mask.data_ptr(), 2.0, k * 256, hidden.data_ptr(), hip.stream_ptr()),
                  "me_mask_scale_f32")
        regress = _linear(hidden, 256, k, 256, w1, b1, LINEAR, _f32(dev, k, 4), 4)
        refine = _linear(hidden, 256, k, 256, w2, b2, SIGMOID, _f32(dev, k, c1), c1)
        x2 = _f32(dev, k * c1, 2)                                                      # [K*(C+1), 2] = (refine, yolo_vector) pairs
        hip.check(lib.me_m2_pairs_f32(refine.data_ptr(), boxes.data_ptr(), cols, k, c1, x2.data_ptr(), hip.stream_ptr()),
                  "me_m2_pairs_f32")
        h1 = _linear(x2, 2, k * c1, 2, e1w, e1b, LEAKY, _f32(dev, k * c1, 32), 32)      # -> flatten [K, 32*(C+1)]
        o = _linear(h1, 32 * c1, k, 32 * c1, e2w, e2b, LEAKY, _f32(dev, k, 2), 2)

        # masks = softmax(o), output rows (reference :341-364): one launch + the compaction / ordering launch of stage 3
        masks, rows_all, key = _f32(dev, k, 2), _f32(dev, k, 8), _f32(dev, k)
        keep = torch.empty((k,), device=dev, dtype=torch.uint8)
        hip.check(lib.me_m2_rows_f32(o.data_ptr(), regress.data_ptr(), boxes.data_ptr(), cols, k, float(net.refine_threshold),
                                     masks.data_ptr(), rows_all.data_ptr(), keep.data_ptr(), key.data_ptr(), hip.stream_ptr()),
                  "me_m2_rows_f32")
        ordered = _f32(dev, k, 8)
        n_out = torch.empty((1,), device=dev, dtype=torch.int32)
        hip.check(lib.me_compact_sort_rows_f32(rows_all.data_ptr(), keep.data_ptr(), key.data_ptr(), k, 8, ordered.data_ptr(),
                                               n_out.data_ptr(), hip.stream_ptr()), "me_compact_sort_rows_f32")
        output = ordered.cpu()[:int(n_out.item())]   # (whole buffers to the host, sliced there: no slicing / casting launches)
        if targets is None:
            # train() mode without targets (reference :299-364 in train mode): batch statistics in fcn_layers (running
            # statistics updated above), Dropout active, no loss
            net._last_train = dict(k=k, refine=refine, masks=masks, boxes=boxes)
            return output

        # labels + sampling on the host (reference :369-420)
        targets[:, 2:] = xywh2xyxy(targets[:, 2:])
        targets[:, 2:] *= size
        # (host-side slicing through numpy: torch's CPU indexing kernels enter an intra-op parallel region, which costs milliseconds on
        #  the many-core hosts of the GPU boxes - 21 ms per step when these three lines were torch ops)
        boxes_np = boxes.cpu().numpy()
        boxes_cpu = torch.from_numpy(np.ascontiguousarray(boxes_np[:, [0, 7, 1, 2, 3, 4]]))
        iou_labels, target_location = iou_labels_vectorized(boxes_cpu, targets.cpu())
        pos_filter = (iou_labels > net.iou_thresh[1]).flatten()
        neg_filter = (iou_labels < net.iou_thresh[0]).flatten()
        flat = iou_labels.flatten()
        conf_1 = torch.from_numpy(np.ascontiguousarray(boxes_np[:, 5]))
        conf_2 = torch.from_numpy(np.ascontiguousarray(masks.cpu().numpy()[:, 1]))
        positive_cpu = torch.from_numpy(keep.cpu().numpy().astype(np.bool_))
        metric = dict(total=len(iou_labels), true=pos_filter.sum(), positive=positive_cpu.sum(),
                      tp=(positive_cpu * pos_filter).sum().float(),
                      conf=dict(conf_1_pos=conf_1[flat > 0.5], conf_1_neg=conf_1[flat < 0.5], conf_2_pos=conf_2[flat > 0.5],
                                conf_2_neg=conf_2[flat < 0.5]))
        pos_idx, neg_idx = np.where(pos_filter)[0], np.where(neg_filter)[0]
        top_k = min(len(pos_idx) * net.balance_fac, len(neg_idx))
        sample_filter = pos_filter.clone()
        sample_filter[neg_idx[random.sample(range(len(neg_idx)), k=top_k)]] = True
        class_label = torch.zeros((k, net.class_num))
        for i, idx in enumerate(pos_idx):  # row i, not idx: the reference's quirk (my_models.py:446-447)
            class_label[i, int(boxes_cpu[idx, 1])] = 1.0
        pos_d = pos_filter.to(torch.uint8).to(dev)
        smp_d = sample_filter.to(torch.uint8).to(dev)
        cl_d, tl_d = class_label.to(dev), target_location.to(dev).contiguous()

        terms = _f32(dev, k, 5)
        d_o, d_ref, d_reg = _f32(dev, k, 2), _f32(dev, k, c1), _f32(dev, k, 4)
        hip.check(lib.me_m2_loss_f32(o.data_ptr(), refine.data_ptr(), c1, regress.data_ptr(), boxes.data_ptr(), cols,
                                     tl_d.data_ptr(), cl_d.data_ptr(), pos_d.data_ptr(), smp_d.data_ptr(), k, float(net.alpha),
                                     float(net.loss_lambda[0]), float(net.loss_lambda[1]), 1.0, terms.data_ptr(),
                                     d_o.data_ptr(), d_ref.data_ptr(), d_reg.data_ptr(), hip.stream_ptr()), "me_m2_loss_f32")
        sums = _f32(dev, 5)
        _colsum(terms, 5, k, 5, sums)
        # focal + (conf + category) / lambda0 + (xy + wh) / lambda1 (reference :456-458) as one [1 x 5] . [5 x 1] product
        loss_val = _f32(dev, 1)
        _gemm(0, 0, 1, 1, 5, sums, 5, _loss_weights(dev, float(net.loss_lambda[0]), float(net.loss_lambda[1])), 1, loss_val, 1)
        loss_val = loss_val.view(())

    names, params = _head_named(net)   # (cached walk: two named_parameters() passes over ~370 tensors were 6 ms of host time per step)
    names, params = list(names), list(params)
    state = dict(loss=loss_val, names=names, net=net, n=n, fh=fh, fw=fw, fc=fc, pix=pix, k=k, c1=c1, ws_t=ws_t, ws=ws, fm=fm,
                 z1=z1, a1=a1, st_img=st_img, rois=rois, feat=feat, t_act=t_act, mask=mask, hidden=hidden, refine=refine, x2=x2,
                 h1=h1, o=o, d_o=d_o, d_ref=d_ref, d_reg=d_reg, w0=w0, w1=w1, w2=w2, e1w=e1w, e2w=e2w, w_img=w_img,
                 terms=sums)
    loss = _StageTwo.apply(state, *params)
    net._last_train = dict(k=k, terms=sums, refine=refine, masks=masks, boxes=boxes)
    return output, loss, metric


def _backward(S, grad_out):
    lib = hip.lib()
    net, k, c1, dev = S["net"], S["k"], S["c1"], S["o"].device
    g = float(grad_out)
    grads = {}
    with torch.no_grad():
        scale = (lambda t: t if g == 1.0 else t * g)
        # ensemble fc2 (LeakyReLU) <- d_o
        dz = _act_bwd(S["o"], scale(S["d_o"]), k, 2, LEAKY)
        dw, db, d_h1 = _linear_bwd(S["h1"], k, 32 * c1, S["e2w"], dz)
        grads["ensemble_head.fc2.0.weight"], grads["ensemble_head.fc2.0.bias"] = dw, db
        # ensemble fc1 over the K*(C+1) (refine, yolo) pairs
        dz = _act_bwd(S["h1"], d_h1.view(k * c1, 32), k * c1, 32, LEAKY)
        dw, db, _none = _linear_bwd(S["x2"], k * c1, 2, S["e1w"], dz, need_dx=False)
        grads["ensemble_head.fc1.0.weight"], grads["ensemble_head.fc1.0.bias"] = dw, db
        # refinement_vector: direct loss gradient + the ensemble path - column 0 of every (refine, yolo) pair: d_x2[:, 0] = dz . e1w[:, 0],
        # accumulated onto the loss gradient by the product itself (n = 1, ldb = 2 walks column 0; the yolo column feeds the frozen
        # detector and is not computed); then the sigmoid
        d_ref = scale(S["d_ref"]).clone()
        _gemm(0, 0, k * c1, 1, 32, dz, 32, S["e1w"], 2, d_ref, 1, beta=1.0)
        dz2 = _act_bwd(S["refine"], d_ref, k, c1, SIGMOID)
        dw, db, d_hid = _linear_bwd(S["hidden"], k, 256, S["w2"], dz2)
        grads["refinement_head.net2.0.weight"], grads["refinement_head.net2.0.bias"] = dw, db
        dw, db, d_hid = _linear_bwd(S["hidden"], k, 256, S["w1"], scale(S["d_reg"]).contiguous(), dx_into=d_hid)
        grads["refinement_head.net1.0.weight"], grads["refinement_head.net1.0.bias"] = dw, db
        # Dropout, LeakyReLU, net0
        d_t = _f32(dev, k, 256)
        hip.check(lib.me_mask_scale_f32(d_hid.data_ptr(), S["mask"].data_ptr(), 2.0, k * 256, d_t.data_ptr(), hip.stream_ptr()),
                  "me_mask_scale_f32")
        dz0 = _act_bwd(S["t_act"], d_t, k, 256, LEAKY)
        dw, db, d_feat = _linear_bwd(S["feat"], k, 490, S["w0"], dz0)
        grads["refinement_head.net0.0.weight"], grads["refinement_head.net0.0.bias"] = dw, db
        # PS-RoIAlign backward -> score map -> BN(train) + leaky -> 1x1 conv
        n, fh, fw, fc, pix = S["n"], S["fh"], S["fw"], S["fc"], S["pix"]
        d_a1 = torch.zeros((pix, 490), device=dev, dtype=torch.float32)
        hip.check(lib.me_ps_roi_align_bwd_f32(d_feat.data_ptr(), S["rois"].data_ptr(), k, n, fh, fw, 490, 7, 1.0 / 16,
                                              d_a1.data_ptr(), 490, hip.stream_ptr()), "me_ps_roi_align_bwd_f32")
        bn = net.fcn_layers.net[1]
        dz1 = _f32(dev, pix, 490)
        dg, dbt = _bn_bwd(S["z1"], 490, d_a1, 490, pix, 490, bn, S["st_img"], LEAKY, dz1, 490, S["ws"])
        grads["fcn_layers.net.batch_norm_0.weight"], grads["fcn_layers.net.batch_norm_0.bias"] = dg, dbt
        dw = _f32(dev, 490, fc)
        _gemm(1, 0, 490, fc, pix, dz1, 490, S["fm"], fc, dw, fc)
        dbc = _f32(dev, 490)
        _colsum(dz1, 490, pix, 490, dbc)
        grads["fcn_layers.net.conv_0.weight"], grads["fcn_layers.net.conv_0.bias"] = dw.view(490, fc, 1, 1), dbc
    for name, p in zip(*_head_named(net)):
        if name in grads and not p.requires_grad:
            grads[name] = None
    return grads
