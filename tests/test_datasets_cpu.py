"""CPU (-m "not gpu"): pin oracle/datasets_ref.py (CPU restatement of the input producer) to the outputs of the
REAL reference MyDataset + collate_fn on the committed mini-dataset (tests/golden/dataset_small.npz), and check the
host-side half of the product's MyDataset (paths, label / radar-box arithmetic) against the same fixture."""
import os

import numpy as np
import torch

from oracle import datasets_ref
from tests.golden.make_golden import DATASET_CASES, DATASET_DIR

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _weighted(imgs, size):
    return (imgs.double() * torch.arange(size, dtype=torch.float64).view(1, 1, 1, -1)
            * torch.arange(1, size + 1, dtype=torch.float64).view(1, 1, -1, 1)).sum((1, 2, 3)).numpy()


def _items(paths):
    out = []
    for rel in paths:
        stem = os.path.splitext(os.path.basename(str(rel)))[0]
        out.append(datasets_ref.load_item(os.path.join(DATASET_DIR, "image", stem + ".jpg"),
                                          os.path.join(DATASET_DIR, "label", stem + ".txt"),
                                          os.path.join(DATASET_DIR, "radar_box", stem + ".pkl"),
                                          os.path.join(DATASET_DIR, "radar_point", stem + ".pkl")))
    return out


def test_oracle_input_producer_matches_reference():
    g = np.load(os.path.join(GOLD, "dataset_small.npz"))
    for mode, size in DATASET_CASES:
        key = f"{mode}{size}"
        imgs, targets, radar_boxes, radar_maps = datasets_ref.collate(_items(g[key + "/paths"]), size, size // 16)
        assert np.array_equal(imgs[:, :, ::3, ::3].numpy(), g[key + "/imgs_sample"])
        assert np.array_equal(imgs.double().sum((1, 2, 3)).numpy(), g[key + "/imgs_sum"])
        assert np.array_equal(_weighted(imgs, size), g[key + "/imgs_wsum"])
        assert np.array_equal(targets.numpy(), g[key + "/targets"])
        assert np.array_equal(radar_boxes.numpy(), g[key + "/radar_boxes"])
        assert np.array_equal(radar_maps.numpy(), g[key + "/radar_maps"])
    assert (g["test416/radar_maps"] > 0).sum() > 20  # the fixture is not trivially empty


def test_product_dataset_host_side_matches_reference():
    """Paths / split, targets and radar boxes are host arithmetic in the product too: exact against the reference."""
    from millieye_amd.utils.datasets import MyDataset, StagedImages, StagedRadarMaps
    g = np.load(os.path.join(GOLD, "dataset_small.npz"))
    for mode, size in DATASET_CASES:
        key = f"{mode}{size}"
        ds = MyDataset(mode=mode, illumination=["H", "L"], img_size=size, augment=False, multiscale=False, test_list=4,
                       dataset_folder=DATASET_DIR)
        paths, imgs, targets, radar_boxes, radar_maps = ds.collate_fn([ds[i] for i in range(len(ds))])
        assert [os.path.relpath(p, DATASET_DIR) for p in paths] == list(g[key + "/paths"])
        assert np.array_equal(targets.numpy(), g[key + "/targets"])
        assert np.array_equal(radar_boxes.numpy(), g[key + "/radar_boxes"])
        assert isinstance(imgs, StagedImages) and tuple(imgs.shape) == (len(ds), 3, size, size)
        assert isinstance(radar_maps, StagedRadarMaps) and tuple(radar_maps.shape) == (len(ds), 3, size // 16, size // 16)
        try:
            imgs.to("cpu")
        except Exception as e:  # no CPU path in the product
            assert "CUDA" in str(e)
        else:
            raise AssertionError("StagedImages.to('cpu') must raise")
    # multiscale: a new size (a multiple of 32 within +-96) is drawn on batches 0, 10, 20 ...
    ds = MyDataset(mode="train", illumination=["H", "L"], img_size=416, multiscale=True, test_list=4, dataset_folder=DATASET_DIR)
    sizes = set()
    for _ in range(12):
        _, imgs, *_rest = ds.collate_fn([ds[0]])
        sizes.add(imgs.size)
    assert all(s % 32 == 0 and 320 <= s <= 512 for s in sizes) and ds.batch_count == 12


def test_oracle_chain_reproduces_reference_evaluate():
    """oracle input producer -> oracle Network.forward -> the product's host-side metrics = the reference's evaluate
    tuple on the mini-dataset (tests/golden/evaluate_small.npz)."""
    from millieye_amd import cfgs
    from millieye_amd.my_models import Network, define_yolo
    from millieye_amd.test_fusion import mode_selection, regroup_outputs
    from millieye_amd.utils.utils import ap_per_class, get_batch_statistics, xywh2xyxy
    from oracle import network_ref
    from tests.golden.make_golden import EVAL_SMALL, eval_small_weights_
    from tests.parity_helpers import cfg_path
    c = EVAL_SMALL
    g = np.load(os.path.join(GOLD, c["name"] + ".npz"))
    d = np.load(os.path.join(GOLD, "dataset_small.npz"))
    net = eval_small_weights_(Network(define_yolo(cfg_path(c["cfg"])), c["conf"]))
    sd, cfg_text = net.state_dict(), cfgs.KNOWN[c["cfg"]]()
    paths = list(d["test416/paths"])
    for model_mode in (0, 3):
        labels, metrics, after = [], [], [1]
        for b0 in range(0, len(paths), c["batch"]):
            imgs, targets, radar_boxes, radar_maps = datasets_ref.collate(_items(paths[b0:b0 + c["batch"]]), 416, 26)
            mode_now = mode_selection(model_mode, imgs, None)
            out = network_ref.network_forward(cfg_text, sd, imgs, radar_maps, radar_boxes, mode_now, c["conf"])
            ref = g[f"mode{model_mode}/rows{b0 // c['batch']}"]
            assert out.shape == ref.shape and np.allclose(out.numpy(), ref, rtol=1e-5, atol=1e-4)
            grouped = regroup_outputs(out, len(imgs))
            after += [len(r) if r is not None else 0 for r in grouped]
            labels += targets[:, 1].tolist()
            targets[:, 2:] = xywh2xyxy(targets[:, 2:])
            targets[:, 2:] *= c["size"]
            metrics += get_batch_statistics(grouped, targets, iou_threshold=0.5)
        tp, conf, pred = [np.concatenate(x, 0) for x in list(zip(*metrics))]
        precision, recall, AP, f1, ap_class, _ = ap_per_class(tp, conf, pred, labels)
        k = f"mode{model_mode}/"
        assert after == list(g[k + "after"]) and list(ap_class) == list(g[k + "ap_class"])
        assert np.allclose(AP, g[k + "AP"]) and np.allclose(precision, g[k + "precision"]) and np.allclose(f1, g[k + "f1"])
