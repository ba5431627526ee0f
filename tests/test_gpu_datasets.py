"""GPU: the device half of the input producer (csrc/input.hip through the C ABI: me_image_pad_resize_u8_f32,
me_radar_heatmap_f32) against the reference's own outputs (tests/golden/dataset_small.npz) and the oracle.
Images (u8 -> /255 -> pad -> nearest) and the heat maps themselves (float64 histograms -> float32) are bit-exact.  The
bilinear resize of the maps is float32 arithmetic whose last bit depends on whether the multiply-adds are fused: the
aten CPU kernel the reference calls is built with FMA contraction, csrc/input.hip rounds every operation separately, so
resized maps agree to 1 ulp (asserted: 1e-6 absolute on values in [0,1]; north_star's bound is 1e-3)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_golden import DATASET_CASES, DATASET_DIR

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _weighted(imgs, size):
    return (imgs.double() * torch.arange(size, dtype=torch.float64).view(1, 1, 1, -1)
            * torch.arange(1, size + 1, dtype=torch.float64).view(1, 1, -1, 1)).sum((1, 2, 3)).numpy()


def test_dataset_batches_match_reference(hip_lib):
    from millieye_amd.utils.datasets import MyDataset
    g = np.load(os.path.join(GOLD, "dataset_small.npz"))
    for mode, size in DATASET_CASES:
        key = f"{mode}{size}"
        ds = MyDataset(mode=mode, illumination=["H", "L"], img_size=size, augment=False, multiscale=False, test_list=4,
                       dataset_folder=DATASET_DIR)
        loader = torch.utils.data.DataLoader(ds, batch_size=len(ds), shuffle=False, num_workers=0, collate_fn=ds.collate_fn)
        (paths, imgs, targets, radar_boxes, radar_maps), = list(loader)
        imgs, radar_maps = imgs.to("cuda"), radar_maps.to("cuda")
        torch.cuda.synchronize()
        assert imgs.is_cuda and imgs.shape == (len(ds), 3, size, size)
        ic, mc = imgs.cpu(), radar_maps.cpu()
        assert np.allclose(ic[:, :, ::3, ::3].numpy(), g[key + "/imgs_sample"], rtol=1e-3, atol=1e-3)
        assert np.allclose(ic.double().sum((1, 2, 3)).numpy(), g[key + "/imgs_sum"], rtol=1e-6)
        assert np.allclose(_weighted(ic, size), g[key + "/imgs_wsum"], rtol=1e-6)  # position-sensitive checksum
        assert np.allclose(mc.numpy(), g[key + "/radar_maps"], rtol=1e-3, atol=1e-3)
        assert np.array_equal(targets.numpy(), g[key + "/targets"])
        assert np.array_equal(radar_boxes.numpy(), g[key + "/radar_boxes"])
        assert np.array_equal(ic[:, :, ::3, ::3].numpy(), g[key + "/imgs_sample"]), key
        assert np.abs(mc.numpy() - g[key + "/radar_maps"]).max() <= 1e-6, key


def test_image_kernel_shapes_vs_oracle(hip_lib):
    """Portrait / landscape / square, up- and down-scaling, identity size - against oracle/datasets_ref.py."""
    from millieye_amd import synth
    from millieye_amd.utils.datasets import StagedImages
    from oracle import datasets_ref
    for tag, (h, w), size in [("a", (37, 91), 64), ("b", (91, 37), 64), ("c", (64, 64), 64), ("d", (50, 50), 200),
                              ("e", (300, 420), 96), ("f", (33, 32), 33), ("g", (450, 800), 416)]:
        frame = torch.from_numpy((synth.uniform("imgk/" + tag, (h, w, 3), 0, 256)).astype(np.uint8))
        got = StagedImages([frame], size).to("cuda").cpu()[0]
        ref = datasets_ref.resize(datasets_ref.pad_to_square(datasets_ref.to_tensor(frame.numpy()), 0)[0], size)
        assert torch.equal(got, ref), (tag, float((got - ref).abs().max()))


def test_radar_heatmap_kernel_vs_oracle(hip_lib):
    """Random point clouds incl. empty frames, points on bin edges / image borders / outside, several per bin."""
    from millieye_amd import synth
    from millieye_amd.utils.datasets import StagedRadarMaps
    from oracle import datasets_ref
    import torch.nn.functional as F
    pts, sizes = [], []
    for i, (w, h, n) in enumerate([(1600, 900, 25), (900, 1600, 60), (640, 640, 0), (1280, 720, 300), (64, 48, 5)]):
        u = synth.uniform(f"hm/{i}/u", (n,), -20, w + 20).astype(np.float64)
        v = synth.uniform(f"hm/{i}/v", (n,), -20, h + 20).astype(np.float64)
        d = synth.uniform(f"hm/{i}/d", (n,), 0.1, 15).astype(np.float64)
        s = synth.uniform(f"hm/{i}/s", (n,), -6, 6).astype(np.float64)
        if n >= 5:
            u[:4], v[:4] = [0.0, w, w / 2.0, w / 4.0], [0.0, h, h / 2.0, h / 4.0]
        pts.append(np.stack([u, v, d, s], 1))
        sizes.append((w, h))
    for ms in (26, 10, 32):
        got = StagedRadarMaps(pts, sizes, ms).to("cuda").cpu()
        for i, (p, (w, h)) in enumerate(zip(pts, sizes)):
            m = datasets_ref.to_tensor(datasets_ref.plot_radar_heatmap(p.transpose(), (w, h))).float()
            m, _ = datasets_ref.pad_to_square(m, 0)
            ref = F.interpolate(m.unsqueeze(0), ms, mode="bilinear", align_corners=True).squeeze(0)
            err = float((got[i] - ref).abs().max())
            if ms == 32:  # the padded map is already 32 x 32: no interpolation arithmetic, the heat map itself
                assert torch.equal(got[i], ref), (ms, i, err)
            else:
                assert err <= 1e-6, (ms, i, err)


def test_evaluate_end_to_end_matches_reference(hip_lib):
    """millieye_amd.test_fusion.evaluate building its own MyDataset (frames decoded by PIL, batches assembled by
    csrc/input.hip, Network.forward on the HIP path, host metrics) against the REAL reference evaluate on the same
    mini-dataset (tests/golden/evaluate_small.npz): per-batch output rows within 1e-3, metric tuples equal."""
    from millieye_amd.my_models import Network, define_yolo
    from millieye_amd.test_fusion import evaluate
    from tests.golden.make_golden import EVAL_SMALL, eval_small_weights_
    from tests.parity_helpers import cfg_path
    c = EVAL_SMALL
    g = np.load(os.path.join(GOLD, c["name"] + ".npz"))
    net = eval_small_weights_(Network(define_yolo(cfg_path(c["cfg"])), c["conf"]))
    net = net.to(net.device)
    rows = []
    real_forward = net.forward

    def fwd(*a, **kw):
        out = real_forward(*a, **kw)
        rows.append(out.detach().cpu().numpy())
        return out

    net.forward = fwd
    for model_mode in (0, 3):
        rows.clear()
        precision, recall, AP, f1, ap_class, box_stat, _pr = evaluate(
            net, mode="test", model_mode=model_mode, illumination=["H", "L"], iou_thresh=0.5, nms_thresh=0.5,
            img_size=c["size"], batch_size=c["batch"], test_list=c["test_list"], dataset_folder=DATASET_DIR, num_workers=0)
        k = f"mode{model_mode}/"
        assert list(box_stat["after"]) == list(g[k + "after"])
        assert list(ap_class) == list(g[k + "ap_class"])
        for name, got in (("precision", precision), ("recall", recall), ("AP", AP), ("f1", f1)):
            assert np.allclose(got, g[k + name], rtol=0, atol=1e-9), (model_mode, name, got, g[k + name])
        assert AP[0] > 0
        for i, r in enumerate(rows):
            ref = g[k + f"rows{i}"]
            assert r.shape == ref.shape, (model_mode, i, r.shape, ref.shape)
            assert np.all(np.abs(r - ref) <= 1e-3 * np.maximum(1.0, np.abs(ref))), (model_mode, i, np.abs(r - ref).max())


def test_batch_statistics_kernel_vs_host(hip_lib):
    """me_batch_statistics_f32 against the host get_batch_statistics (itself pinned bit-for-bit to the reference,
    tests/golden/metrics_synth.npz): random detections around the targets, exact IoU ties, duplicate boxes, labels
    absent from the targets, images without targets / without detections, early exit once all targets are taken."""
    from millieye_amd import synth
    from millieye_amd.test_fusion import batch_statistics_device, regroup_outputs
    from millieye_amd.utils.utils import get_batch_statistics
    n_img, q, m = 6, 23, 400
    t_xy = synth.uniform("bs/txy", (q, 2), 20, 300)
    t_wh = synth.uniform("bs/twh", (q, 2), 20, 120)
    t_img = np.floor(synth.uniform("bs/ti", (q, 1), 0, 4))          # images 4, 5 have no targets
    t_lab = np.floor(synth.uniform("bs/tl", (q, 1), 0, 2))
    targets = np.concatenate([t_img, t_lab, t_xy, t_xy + t_wh], 1).astype(np.float32)
    targets[7, 2:] = targets[6, 2:]                                 # duplicate target box (tie -> first index)
    targets[7, 0] = targets[6, 0]
    pick = np.floor(synth.uniform("bs/pick", (m,), 0, q)).astype(int)
    jitter = synth.uniform("bs/jit", (m, 4), -12, 12)
    boxes = targets[pick, 2:] + jitter
    boxes[:40] = targets[pick[:40], 2:]                             # exact copies: IoU 1.0, ties between duplicates
    img = targets[pick, 0].copy()
    img[350:] = np.floor(synth.uniform("bs/oi", (50,), 3, 6))       # detections on images 3 (no dets otherwise), 4, 5
    p = synth.uniform("bs/p", (m,), 0.01, 0.99)
    lab = np.floor(synth.uniform("bs/l", (m,), 0, 3))               # label 2 never occurs among the targets
    order = np.argsort(-p, kind="stable")
    rows = np.concatenate([img[:, None], boxes, p[:, None], p[:, None] * 0.5, lab[:, None]], 1).astype(np.float32)[order]
    out = torch.from_numpy(rows).cuda()
    tg = torch.from_numpy(targets)
    for thr in (0.5, 0.3, 0.9):
        got = batch_statistics_device(out, tg, n_img, thr)
        ref = get_batch_statistics(regroup_outputs(out, n_img), tg, iou_threshold=thr)
        assert len(got) == len(ref) and len(ref) >= 4
        n_tp = 0
        for (tp_g, sc_g, lb_g), (tp_r, sc_r, lb_r) in zip(got, ref):
            assert np.array_equal(tp_g, np.asarray(tp_r)), thr
            assert np.array_equal(sc_g, np.asarray(sc_r)) and np.array_equal(lb_g, np.asarray(lb_r))
            n_tp += int(tp_g.sum())
        assert n_tp > 5
    assert batch_statistics_device(out[:0], tg, n_img, 0.5) == []
    none = batch_statistics_device(out, tg[:0], n_img, 0.5)         # no targets at all: every TP is 0
    assert sum(int(x[0].sum()) for x in none) == 0 and len(none) == len(ref)


@pytest.mark.parametrize("dtype,ap_tol", [("bf16", 0.03), ("f16", 0.01)])
def test_evaluate_in_16bit_storage_modes(hip_lib, dtype, ap_tol):
    """mAP@0.5 of the whole evaluation chain with the detector in a 16-bit storage mode against the REAL reference's fp32
    numbers on the mini-dataset (tests/golden/evaluate_small.npz): same detected classes, AP / precision / recall within
    ``ap_tol`` absolute (bf16: 3 points, IEEE half: 1 point) - the storage error moves a few boxes across the IoU / confidence
    thresholds, it must not change the picture."""
    from millieye_amd.my_models import Network, define_yolo
    from millieye_amd.test_fusion import evaluate
    from tests.golden.make_golden import EVAL_SMALL, eval_small_weights_
    from tests.parity_helpers import cfg_path
    c = EVAL_SMALL
    g = np.load(os.path.join(GOLD, c["name"] + ".npz"))
    net = eval_small_weights_(Network(define_yolo(cfg_path(c["cfg"])), c["conf"]))
    net = net.to(net.device)
    net.base_detector.compute_dtype = dtype
    for model_mode in (0, 3):
        precision, recall, AP, f1, ap_class, box_stat, _pr = evaluate(
            net, mode="test", model_mode=model_mode, illumination=["H", "L"], iou_thresh=0.5, nms_thresh=0.5,
            img_size=c["size"], batch_size=c["batch"], test_list=c["test_list"], dataset_folder=DATASET_DIR, num_workers=0)
        k = f"mode{model_mode}/"
        assert list(ap_class) == list(g[k + "ap_class"])
        for name, got in (("precision", precision), ("recall", recall), ("AP", AP)):
            assert np.all(np.abs(np.asarray(got) - g[k + name]) <= ap_tol), (dtype, model_mode, name, got, g[k + name])
        print(f"{dtype} mode {model_mode}: AP {np.asarray(AP)} (reference fp32 {g[k + 'AP']})")
