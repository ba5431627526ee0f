"""One camera frame + its radar frames through the whole fusion path, the way the live demos do it
(``module3_our_dataset/run_mp.py:65-160,296-330`` / ``run_sp.py:117-241``) - without the demos' I/O (video decoding,
serial-port capture, OpenCV window).  ``millieye_amd/pipeline.py`` runs the two halves of
:class:`FrameFuser` in two processes with run_mp's ``mp.Queue(maxsize=3)`` / ``mp.Event`` hand-over:

    radar frames --RadarProposalGenerator--> box proposals (pixels) + point cloud
    frame uint8 [h,w,3] --ToTensor / pad_to_square / resize(416)--> img [1,3,416,416]     (me_image_pad_resize_u8_f32)
    point cloud --plot_radar_heatmap / pad_to_square--> radar_map [1,3,32,32]              (me_radar_heatmap_f32, no resize:
                                                                                            the demos feed the raw map, quirk q15)
    proposals --+pad, /padded side, clamp, drop empty--> radar_box [k,5]                   (run_mp.py:120-135)
    mode: auto = fusion iff img.mean() < 0.08, else camera only                            (run_mp.py:204-212)
    Network.forward(img, radar_map, radar_box, mode)[:, 1:] -> batched_nms(.., 0.3) -> rescale_boxes to the frame

Parity: every numeric stage is one of the pinned pieces (input kernels: tests/golden/dataset_small; Network.forward:
network_*.npz; NMS: nms_synth; radar proposals: radar_proposals_synth + the unpinned Kalman part); the glue in this file
restates run_mp's inline code and has no fixture of its own.
"""
import numpy as np
import torch

from . import hip
from .radar_proposals import RadarProposalGenerator
from .utils.datasets import StagedImages, StagedRadarMaps, _pad_amounts
from .utils.utils import box_ops, rescale_boxes

__all__ = ["mode_selection", "radar_boxes_for_network", "FrameFuser"]


def mode_selection(mode, img, dark_threshold=0.08):
    """0 milliEye fusion, 1 camera only, 2 radar only, 3 auto: fusion when the frame is dark (run_mp.py:204-212; the offline
    evaluation uses 0.1, test_fusion.py:24-32)."""
    if mode in (0, 1, 2):
        return mode
    if mode == 3:
        return 0 if float(img.mean()) < dark_threshold else 1
    return None  # the reference falls off the end for other values


def radar_boxes_for_network(xyxy_pixels, frame_hw):
    """Pixel boxes of the un-padded frame -> ``[k,5]`` rows ``(0, x1, y1, x2, y2)`` in units of the padded square side,
    clamped to [0,1], empty boxes dropped (run_mp.py:119-135)."""
    h, w = frame_hw
    boxes = torch.as_tensor(np.asarray(xyxy_pixels, dtype=np.float32).reshape(-1, 4))
    if len(boxes) == 0:
        return torch.zeros((0, 5))
    left, right, top, bottom = _pad_amounts(h, w)
    side = float(max(h, w))
    boxes = boxes + torch.tensor([left, top, right, bottom], dtype=torch.float32)
    boxes = torch.clamp(boxes / side, 0, 1)
    boxes = boxes[(boxes[:, 0] < boxes[:, 2]) & (boxes[:, 1] < boxes[:, 3])]
    out = torch.zeros((len(boxes), 5))
    out[:, 1:] = boxes
    return out


class FrameFuser:
    """``fuser(frame_uint8_hwc, radar_frames)`` -> ``(detections [m,7], info)``; detections are rows
    ``(x1, y1, x2, y2, conf, cls_conf, cls_pred)`` in pixels of the original frame, like the demos draw them."""

    def __init__(self, model, calib_param, model_mode=3, img_size=416, nms_iou=0.3, dark_threshold=0.08, generator=None,
                 **generator_kwargs):
        self.model, self.model_mode, self.img_size = model, model_mode, img_size
        self.nms_iou, self.dark_threshold = nms_iou, dark_threshold
        self.generator_is_default = generator is None   # pipeline.py: re-created in the producer vs sent by pickle
        self.generator = generator or RadarProposalGenerator(calib_param, **generator_kwargs)

    def __call__(self, frame, radar_frames):
        return self.infer(self.prepare(frame, radar_frames))

    # The two halves run_mp.py puts in two processes (millieye_amd/pipeline.py does the same): everything that needs only
    # the host - radar tracking, proposal arithmetic, staging of the raw frame bytes / point cloud - and everything that
    # needs the device.  ``prepare`` keeps the tracker state, so it must see the frames in order.
    def prepare(self, frame, radar_frames):
        """Host half (run_mp.py:65-152 ``pre_process``): a picklable payload for :meth:`infer`."""
        frame = torch.as_tensor(frame)
        if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
            raise hip.MeError(f"frame must be uint8 [h,w,3] (got {frame.dtype} {tuple(frame.shape)})")
        h, w = int(frame.shape[0]), int(frame.shape[1])
        proposals, cloud = self.generator(radar_frames)
        return dict(hw=(h, w), proposals=proposals, points=int(len(cloud)),
                    radar_box=radar_boxes_for_network(proposals, (h, w)),
                    img=StagedImages([frame], self.img_size), radar_map=StagedRadarMaps([cloud], [(w, h)], map_size=32))

    def infer(self, payload):
        """Device half (run_mp.py:296-330): input kernels, mode selection, ``Network.forward``, second NMS, rescale."""
        dev = getattr(self.model, "device", None) or hip.default_device()
        h, w = payload["hw"]
        radar_box = payload["radar_box"].to(dev)
        img = payload["img"].to(dev)
        radar_map = payload["radar_map"].to(dev)
        mode = mode_selection(self.model_mode, img, self.dark_threshold)
        with torch.no_grad():
            rows = self.model(img, radar_map, radar_box, mode)[:, 1:].cpu()
        keep = box_ops.batched_nms(rows[:, :4], rows[:, 4], rows[:, 6], self.nms_iou)
        rows = rows[keep]
        if len(rows):
            rescale_boxes(rows, self.img_size, (h, w))
        return rows, dict(mode=mode, proposals=payload["proposals"], radar_boxes=int(radar_box.shape[0]),
                          points=payload["points"])
