"""CPU (-m "not gpu"): pin the ORACLE's training step (oracle/network_ref.network_train_step) to the
reference's own training forward/backward (tests/golden/train_*.npz), and the data-parallel gradient
bucket helper with a world_size-2 gloo run."""
import os
import random

import numpy as np
import torch

from millieye_amd import cfgs, synth
from oracle import network_ref
from tests.golden.make_golden import TRAIN_CASE, train_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_training_step_matches_reference():
    from millieye_amd.my_models import Network, define_yolo
    from tests.parity_helpers import cfg_path
    name, cfg, n, s, conf, seed = TRAIN_CASE
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = Network(define_yolo(cfg_path(cfg)), conf)
    synth.fill_network_(net, name)
    x, maps, rboxes = train_inputs(name, n, s)
    targets = torch.from_numpy(g["targets"])
    random.seed(seed)
    res = network_ref.network_train_step(cfgs.KNOWN[cfg](), net.state_dict(), x, maps, rboxes, targets, conf_thresh=conf)
    assert abs(float(res["loss"]) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    assert res["n_pos"] == int(g["n_pos"]) and res["n_pos"] > 0
    assert np.allclose(res["output"].numpy(), g["output"], rtol=1e-5, atol=1e-5)
    seen = 0
    for key in g.files:
        if key.startswith("gnorm/"):
            k = key[6:]
            gr = res["grads"][k]
            assert abs(float(gr.double().norm()) - float(g[key])) <= 1e-4 * max(1e-6, float(g[key])), k
            samp = gr.flatten()[::max(1, gr.numel() // 64)].numpy()
            assert np.allclose(samp, g["gsamp/" + k], rtol=1e-4, atol=1e-6), k
            seen += 1
        elif key.startswith("buf/"):
            assert np.allclose(res["buffers"][key[4:]].numpy(), g[key], rtol=1e-5, atol=1e-6), key
    assert seen >= 20  # every trainable head tensor that the loss reaches
    for k in ("refinement_head.net1.0.weight", "refinement_head.net3.0.weight", "refinement_head.fusion_head.0.weight"):
        assert res["grads"][k] is None  # regression loss excluded / layers unused


def test_oracle_eval_mode_training_step_matches_reference():
    """The same call on a model left in eval() mode (the reference has no mode check, my_models.py:545-641): the oracle's
    ``bn_training=False`` step against the real reference's eval-mode run (train_tiny12_s160_n2_bneval.npz): loss, rows,
    every gradient - the conv biases in front of the BatchNorms now receive real gradients - and untouched running
    statistics."""
    from millieye_amd.my_models import Network, define_yolo
    from tests.golden.make_golden import TRAIN_EVAL_NAME
    from tests.parity_helpers import cfg_path
    name, cfg, n, s, conf, seed = TRAIN_CASE
    g = np.load(os.path.join(GOLD, TRAIN_EVAL_NAME + ".npz"))
    net = Network(define_yolo(cfg_path(cfg)), conf)
    synth.fill_network_(net, name)
    sd = net.state_dict()
    x, maps, rboxes = train_inputs(name, n, s)
    random.seed(seed)
    res = network_ref.network_train_step(cfgs.KNOWN[cfg](), sd, x, maps, rboxes, torch.from_numpy(g["targets"]),
                                         conf_thresh=conf, bn_training=False)
    assert abs(float(res["loss"]) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    assert res["n_pos"] == int(g["n_pos"]) and res["n_pos"] > 0
    assert np.allclose(res["output"].numpy(), g["output"], rtol=1e-5, atol=1e-5)
    seen = 0
    for key in g.files:
        if key.startswith("gnorm/"):
            k = key[6:]
            gr = res["grads"][k]
            assert abs(float(gr.double().norm()) - float(g[key])) <= 1e-4 * max(1e-6, float(g[key])), k
            seen += 1
        elif key.startswith("buf/"):
            assert np.array_equal(g[key], sd[key[4:]].numpy()), f"{key}: eval mode must not update running statistics"
    assert seen >= 20
    assert float(g["gnorm/img_cnn_layers.net.conv_0.bias"]) > 1e-6


def test_vectorised_iou_labels_equal_the_reference_loop():
    """train_path.iou_labels_vectorized must reproduce the reference's per-box loop bit for bit
    (ties, empty filters, NaN-free inputs), incl. against the oracle's own restatement."""
    from millieye_amd.my_models import obtain_iou_labels
    from millieye_amd.train_path import iou_labels_vectorized
    P, Q = 300, 17
    xy = synth.uniform("il/xy", (P, 2), 0, 300)
    wh = synth.uniform("il/wh", (P, 2), 5, 120)
    boxes = np.concatenate([np.floor(synth.uniform("il/i", (P, 1), 0, 4)), np.floor(synth.uniform("il/c", (P, 1), 0, 2)),
                            xy, xy + wh], 1).astype(np.float32)
    t_xy = synth.uniform("il/txy", (Q, 2), 0, 300)
    t_wh = synth.uniform("il/twh", (Q, 2), 5, 120)
    targets = np.concatenate([np.floor(synth.uniform("il/ti", (Q, 1), 0, 3)), np.zeros((Q, 1), np.float32),
                              t_xy, t_xy + t_wh], 1).astype(np.float32)
    targets[5] = targets[4]  # an exact tie: the first of the two must win
    boxes[:10, 2:6] = targets[:10, 2:6]
    boxes[:10, :2] = targets[:10, :2]
    b, t = torch.from_numpy(boxes), torch.from_numpy(targets)
    l0, loc0 = obtain_iou_labels(b, t, (0.3, 0.7))
    l1, loc1 = iou_labels_vectorized(b, t)
    l2, loc2 = network_ref.obtain_iou_labels(b, t, (0.3, 0.7))
    assert torch.equal(l0, l1) and torch.equal(loc0, loc1)
    assert torch.equal(l0, l2) and torch.equal(loc0, loc2)
    assert (l0 > 0.7).sum() >= 10 and (l0 == 0).sum() > 0
    e, _ = iou_labels_vectorized(b[:0], t)
    assert e.shape == (0, 1)


def test_oracle_detector_training_step_matches_reference():
    """oracle/darknet_ref.darknet_train_step (loss + gradient of every detector parameter, eval-mode BatchNorm) against
    the reference's own autograd run (tests/golden/yololoss_tiny12_s96_n2.npz)."""
    from oracle import darknet_ref
    from tests.golden.make_golden import YOLO_LOSS_CASE
    from tests.parity_helpers import make_darknet
    name, cfg, n, s = YOLO_LOSS_CASE
    g = np.load(os.path.join(GOLD, name + ".npz"))
    model = make_darknet(cfg, name)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    loss, grads = darknet_ref.darknet_train_step(cfgs.KNOWN[cfg](), model.state_dict(), x, torch.from_numpy(g["targets"]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    checked = 0
    for key in g.files:
        if key.startswith("gnorm/"):
            k = key[6:]
            gr = grads[k]
            assert abs(float(gr.double().norm()) - float(g[key])) <= 1e-4 * max(1e-6, float(g[key])), k
            assert np.allclose(gr.flatten()[::max(1, gr.numel() // 64)].numpy(), g["gsamp/" + k], rtol=1e-4, atol=1e-6), k
            checked += 1
    assert checked == 37


def test_oracle_yolo_loss_terms_match_the_reference_metrics():
    """oracle/darknet_ref.yolo_loss_terms (the checker of the device loss kernel me_yolo_loss_fwd_f32: loss terms, the
    reference's metrics dict, dense build_targets tensors) on the oracle's own raw detection maps against the numbers the real
    reference's YOLO layers reported (tests/golden/yololoss_tiny12_s96_n2.npz: total loss and m{i}/{metric} per scale)."""
    from oracle import darknet_ref
    from tests.golden.make_golden import YOLO_LOSS_CASE
    from tests.parity_helpers import make_darknet
    name, cfg, n, s = YOLO_LOSS_CASE
    g = np.load(os.path.join(GOLD, name + ".npz"))
    model = make_darknet(cfg, name)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    text = cfgs.KNOWN[cfg]()
    _, _, outs = darknet_ref.darknet_forward(text, model.state_dict(), x, return_layers=True)
    blocks = darknet_ref.parse_cfg_text(text)[1:]
    targets = torch.from_numpy(g["targets"])
    total, scale = 0.0, 0
    for i, b in enumerate(blocks):
        if b["type"] != "yolo":
            continue
        raw = outs[i - 1].permute(0, 2, 3, 1).contiguous()  # the detection conv's output, NHWC
        loss, metrics, dense = darknet_ref.yolo_loss_terms(raw, darknet_ref._anchors_of(b), int(b["classes"]), s, targets.clone())
        for k, v in metrics.items():
            ref = float(g[f"m{scale}/{k}"])
            assert abs(float(v) - ref) <= 1e-5 * max(1.0, abs(ref)), (scale, k, v, ref)
        assert dense["n_obj"] > 0 and dense["tcls"].sum() >= dense["n_obj"]
        total += float(loss)
        scale += 1
    assert scale == 2 and abs(total - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))


def test_oracle_detector_training_step_train_mode_bn_matches_reference():
    """Same with the detector in train() mode: batch-statistics BatchNorm (momentum 0.9), loss, gradients and the
    updated running statistics against the reference's own run (yololoss_tiny12_s96_n2_bntrain.npz)."""
    from oracle import darknet_ref
    from tests.golden.make_golden import YOLO_LOSS_CASE
    from tests.parity_helpers import make_darknet
    name, cfg, n, s = YOLO_LOSS_CASE
    g = np.load(os.path.join(GOLD, name + "_bntrain.npz"))
    model = make_darknet(cfg, name)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    loss, grads, bufs = darknet_ref.darknet_train_step(cfgs.KNOWN[cfg](), model.state_dict(), x,
                                                       torch.from_numpy(g["targets"]), training=True)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for key in g.files:
        if key.startswith("gnorm/"):
            k = key[6:]
            assert abs(float(grads[k].double().norm()) - float(g[key])) <= 2e-4 * max(1e-6, float(g[key])), k
        elif key.startswith("buf/"):
            assert np.allclose(bufs[key[4:]].numpy(), g[key], rtol=1e-5, atol=1e-6), key
