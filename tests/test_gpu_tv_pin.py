"""-m gpu: pin the torchvision boundary to an INSTALLED torchvision, if the box has one.

The reference calls three third-party operators - ``torchvision.ops.ps_roi_align`` / ``roi_align``
(module3_our_dataset/my_models.py:495-496) and ``torchvision.ops.boxes.batched_nms`` (module3_our_dataset/utils/utils.py:372).
torchvision is in neither the build container nor the GPU box of this pool (profiles/r06_torchvision_probe.txt:
``ModuleNotFoundError`` on both, no wheel on disk), so the whole module is skipped there and the three rows stay
"parity unpinned" (DESIGN.md §4).  On any machine that does have the library this file is the pin: it compares BOTH the
oracle's restatement (oracle/tv_ops.c) AND the HIP kernels with the library's CPU operators - kept indices bit-exact,
pooled values to 1e-6 of the map's range (the library accumulates in another order), both backward operators through autograd.
Nothing of /root/reference is needed at run time: torchvision is a third-party library, not the reference."""
import numpy as np
import pytest
import torch

tv = pytest.importorskip("torchvision", reason="torchvision is not installed on this box (profiles/r06_torchvision_probe.txt)")
from torchvision import ops as tvo  # noqa: E402

from millieye_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu
VAL_TOL = 1e-6


def _rois(tag, k, n, size, degenerate=True):
    c = synth.uniform(tag + "c", (k, 2), 0.1 * size, 0.9 * size)
    half = synth.uniform(tag + "h", (k, 2), 2.0, 0.45 * size)
    idx = np.floor(synth.uniform(tag + "i", (k, 1), 0, n))
    r = np.concatenate([idx, c - half, c + half], 1).astype(np.float32)
    if degenerate:
        r[0, 1:] = [10.0, 10.0, 10.0, 10.0]                    # zero-area box
        r[1, 1:] = [-50.0, -30.0, size + 80.0, size + 40.0]    # larger than the map
        r[2, 1:] = [100.0, 90.0, 60.0, 40.0]                   # inverted box
    return torch.from_numpy(r)


def _same(got, ref, what, tol=VAL_TOL):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, what
    nan = ref.isnan()
    assert torch.equal(got.isnan(), nan), f"{what}: NaN pattern differs"
    inf = ref.isinf()
    assert torch.equal(got.isinf(), inf) and torch.equal(got[inf], ref[inf]), f"{what}: inf pattern differs"
    ok = ~(nan | inf)
    if ok.any():
        scale = max(1.0, float(ref[ok].abs().max()))
        err = float((got[ok] - ref[ok]).abs().max()) / scale
        assert err <= tol, f"{what}: {err:.3e} > {tol:.1e}"


@pytest.mark.parametrize("h,w,n,k", [(26, 26, 2, 40), (10, 10, 3, 40), (13, 20, 1, 40), (26, 26, 8, 2400)])
def test_roi_forward_oracle_and_hip_vs_torchvision(hip_lib, h, w, n, k):
    from millieye_amd import hip
    from oracle import tv_ops
    size = 16.0 * max(h, w)
    rois = _rois(f"pin{h}{k}", k, n, size)
    m10 = torch.from_numpy(synth.uniform(f"pin10{h}", (n, 10, h, w), -1, 1))
    m490 = torch.from_numpy(synth.uniform(f"pin490{h}", (n, 490, h, w), -1, 1))
    lib_r = tvo.roi_align(m10, rois, (7, 7), 1 / 16, sampling_ratio=-1, aligned=False)
    lib_p = tvo.ps_roi_align(m490, rois, (7, 7), 1 / 16, sampling_ratio=-1)
    _same(tv_ops.roi_align(m10, rois, (7, 7), 1 / 16), lib_r, "oracle roi_align")
    _same(tv_ops.ps_roi_align(m490, rois, (7, 7), 1 / 16), lib_p, "oracle ps_roi_align")
    _same(hip.roi_align(m10.permute(0, 2, 3, 1).contiguous().cuda(), rois), lib_r, "HIP roi_align")
    _same(hip.ps_roi_align(m490.permute(0, 2, 3, 1).contiguous().cuda(), rois), lib_p, "HIP ps_roi_align")


def test_roi_backward_oracle_and_hip_vs_torchvision(hip_lib):
    from millieye_amd import hip
    from oracle import tv_ops
    n, h, w, k = 2, 10, 10, 24
    rois = _rois("pinb", k, n, 160.0, degenerate=False)
    for ps, ch in ((False, 10), (True, 490)):
        base = torch.from_numpy(synth.uniform(f"pinbm{ch}", (n, ch, h, w), -1, 1))
        g = None
        grads = []
        for fn in ((tvo.ps_roi_align if ps else tvo.roi_align), (tv_ops.ps_roi_align if ps else tv_ops.roi_align)):
            m = base.clone().requires_grad_(True)
            out = fn(m, rois, (7, 7), 1 / 16)
            if g is None:
                g = torch.from_numpy(synth.uniform(f"pinbg{ch}", tuple(out.shape), -1, 1))
            out.backward(g)
            grads.append(m.grad)
        _same(grads[1], grads[0], f"oracle {'ps_' if ps else ''}roi_align backward", 1e-5)
        gmap = torch.zeros((n, h, w, ch)).cuda()
        gd, rd = g.cuda(), rois.cuda()
        fn = hip.lib().me_ps_roi_align_bwd_f32 if ps else hip.lib().me_roi_align_bwd_f32
        hip.check(fn(gd.data_ptr(), rd.data_ptr(), k, n, h, w, ch, 7, 1.0 / 16, gmap.data_ptr(), ch, hip.stream_ptr()), "roi bwd")
        torch.cuda.synchronize()
        _same(gmap.permute(0, 3, 1, 2), grads[0], f"HIP {'ps_' if ps else ''}roi_align backward", 1e-5)


@pytest.mark.parametrize("m,classes,thr", [(0, 1, 0.5), (1, 1, 0.5), (300, 12, 0.5), (2535, 12, 0.5), (5000, 3, 0.3), (22743, 80, 0.5)])
def test_batched_nms_oracle_and_hip_vs_torchvision(hip_lib, m, classes, thr):
    from millieye_amd import hip
    from oracle import tv_ops
    g = np.random.RandomState(m + classes)
    a = g.uniform(0, 416, size=(m, 2)).astype(np.float32)
    wh = g.uniform(4, 160, size=(m, 2)).astype(np.float32)
    boxes = np.concatenate([a, a + wh], 1)
    if m > 6:
        boxes[3] = boxes[2]                                   # identical boxes: IoU exactly 1
        boxes[5] = boxes[4] + np.float32(0.25)
    scores = torch.from_numpy(g.permutation(m).astype(np.float32) / max(m, 1))  # distinct scores: the visiting order is defined
    idxs = torch.from_numpy(g.randint(0, classes, size=m).astype(np.float32))
    boxes = torch.from_numpy(boxes)
    ref = tvo.boxes.batched_nms(boxes, scores, idxs.long(), thr)
    assert tv_ops.batched_nms(boxes, scores, idxs, thr).tolist() == ref.tolist(), "oracle batched_nms"
    got = hip.nms_indices(boxes.cuda(), scores.cuda(), idxs.cuda(), thr).cpu()
    assert got.tolist() == ref.tolist(), "HIP batched_nms"
    ref1 = tvo.nms(boxes, scores, thr)
    assert tv_ops.nms(boxes, scores, thr).tolist() == ref1.tolist(), "oracle nms"
