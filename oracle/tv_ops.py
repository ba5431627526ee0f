"""ORACLE (test infrastructure): Python face of oracle/tv_ops.c - torch-tensor wrappers with the
call signatures of ``torchvision.ops.roi_align`` / ``ps_roi_align`` / ``boxes.nms`` /
``boxes.batched_nms`` (autograd-capable, so the imported reference can run its training tail).

PARITY UNPINNED - see the header of tv_ops.c.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmillieye_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "tv_ops.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libmillieye_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        F, I, P = C.c_float, C.c_int, C.c_void_p
        _lib.tv_roi_align_forward.argtypes = [P, I, I, I, P, I, I, F, I, I, P]
        _lib.tv_roi_align_backward.argtypes = [P, I, I, I, I, P, I, I, F, I, I, P]
        _lib.tv_ps_roi_align_forward.argtypes = [P, I, I, I, P, I, I, F, I, P]
        _lib.tv_ps_roi_align_backward.argtypes = [P, I, I, I, P, I, I, F, I, P]
        _lib.tv_nms.argtypes = [P, P, I, F, P]
        _lib.tv_nms.restype = I
        _lib.tv_batched_nms.argtypes = [P, P, P, I, F, P]
        _lib.tv_batched_nms.restype = I
    return _lib


def _f32(t):
    return t.detach().to("cpu", torch.float32).contiguous()


def _pooled(output_size):
    if isinstance(output_size, int):
        return output_size
    assert output_size[0] == output_size[1], "square pooling only"
    return int(output_size[0])


class _RoIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, rois, pooled, scale, sampling_ratio, aligned):
        x, r = _f32(inp), _f32(rois)
        n, c, h, w = x.shape
        k = r.shape[0]
        out = torch.zeros((k, c, pooled, pooled), dtype=torch.float32)
        if k:
            lib().tv_roi_align_forward(x.data_ptr(), c, h, w, r.data_ptr(), k, pooled, scale, sampling_ratio,
                                       int(aligned), out.data_ptr())
        ctx.save_for_backward(r)
        ctx.meta = (n, c, h, w, pooled, scale, sampling_ratio, int(aligned))
        return out

    @staticmethod
    def backward(ctx, grad):
        (r,) = ctx.saved_tensors
        n, c, h, w, pooled, scale, sr, aligned = ctx.meta
        g = _f32(grad)
        gin = torch.zeros((n, c, h, w), dtype=torch.float32)
        if r.shape[0]:
            lib().tv_roi_align_backward(g.data_ptr(), n, c, h, w, r.data_ptr(), r.shape[0], pooled, scale, sr, aligned,
                                        gin.data_ptr())
        return gin, None, None, None, None, None


class _PSRoIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, rois, pooled, scale, sampling_ratio):
        x, r = _f32(inp), _f32(rois)
        n, c, h, w = x.shape
        assert c % (pooled * pooled) == 0, "input channels must be divisible by pooled area"
        k = r.shape[0]
        out = torch.zeros((k, c // (pooled * pooled), pooled, pooled), dtype=torch.float32)
        if k:
            lib().tv_ps_roi_align_forward(x.data_ptr(), c, h, w, r.data_ptr(), k, pooled, scale, sampling_ratio,
                                          out.data_ptr())
        ctx.save_for_backward(r)
        ctx.meta = (n, c, h, w, pooled, scale, sampling_ratio)
        return out

    @staticmethod
    def backward(ctx, grad):
        (r,) = ctx.saved_tensors
        n, c, h, w, pooled, scale, sr = ctx.meta
        g = _f32(grad)
        gin = torch.zeros((n, c, h, w), dtype=torch.float32)
        if r.shape[0]:
            lib().tv_ps_roi_align_backward(g.data_ptr(), c, h, w, r.data_ptr(), r.shape[0], pooled, scale, sr,
                                           gin.data_ptr())
        return gin, None, None, None, None


def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    return _RoIAlign.apply(input, boxes, _pooled(output_size), float(spatial_scale), int(sampling_ratio), aligned)


def ps_roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1):
    return _PSRoIAlign.apply(input, boxes, _pooled(output_size), float(spatial_scale), int(sampling_ratio))


def nms(boxes, scores, iou_threshold):
    b, s = _f32(boxes), _f32(scores)
    m = b.shape[0]
    keep = torch.empty((m,), dtype=torch.int64)
    nk = lib().tv_nms(b.data_ptr(), s.data_ptr(), m, float(iou_threshold), keep.data_ptr()) if m else 0
    return keep[:nk]


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b, s, i = _f32(boxes), _f32(scores), _f32(idxs)
    m = b.shape[0]
    keep = torch.empty((m,), dtype=torch.int64)
    nk = lib().tv_batched_nms(b.data_ptr(), s.data_ptr(), i.data_ptr(), m, float(iou_threshold), keep.data_ptr())
    return keep[:nk]
