"""Shared builders for the parity tests, smoke() and bench.py: deterministic models / inputs and
the comparison conventions (tolerances are written where they are used)."""
import os
import tempfile

import torch

from millieye_amd import cfgs, synth

CFG_DIR = os.path.join(tempfile.gettempdir(), f"millieye_amd_testcfg_{os.getuid()}")


def cfg_path(name):
    return cfgs.write_cfg(name, CFG_DIR)


def cfg_text(name):
    return cfgs.KNOWN[name]()


def mini32_cfg_text(classes=4, size=64):
    """A 20-module detector whose channel counts are all multiples of 32 (what the 16-bit training step wants) with every block
    kind of yolov3.cfg - stride-2 convolutions, two residual blocks, two scales joined by upsample + route - and a convolution at
    module 8 (the reference's ``conv_8`` feature tap).  Test-only: shallow enough that rounding noise does not compound through 75
    normalisations, so a 16-bit step can be compared tensor by tensor."""
    e = cfgs._Emitter(size)
    det = 3 * (classes + 5)
    e.conv(32, 3)            # 0
    e.conv(64, 3, stride=2)  # 1
    e.conv(32, 1)            # 2
    e.conv(64, 3)            # 3
    e.shortcut(-3)           # 4
    e.conv(128, 3, stride=2) # 5
    e.conv(64, 1)            # 6
    e.conv(128, 3)           # 7
    e.conv(64, 1)            # 8  (tap)
    e.conv(128, 3)           # 9
    e.shortcut(-3)           # 10 (+ module 7)
    e.conv(det, 1, bn=False, act="linear")  # 11
    e.yolo((3, 4, 5), cfgs._TINY_ANCHORS, classes, 6)  # 12
    e.route(-3)              # 13 (= module 10)
    e.conv(64, 1)            # 14
    e.upsample(2)            # 15
    e.route(-1, 4)           # 16: 64 + 64 channels at stride 2
    e.conv(128, 3)           # 17
    e.conv(det, 1, bn=False, act="linear")  # 18
    e.yolo((0, 1, 2), cfgs._TINY_ANCHORS, classes, 6)  # 19
    return e.text()


def make_mini32(tag, size=64):
    """(Darknet module tree on the mini cfg, its cfg text)."""
    from millieye_amd.yolov3.models import Darknet
    text = mini32_cfg_text(size=size)
    os.makedirs(CFG_DIR, exist_ok=True)
    path = os.path.join(CFG_DIR, "mini32.cfg")
    with open(path, "w") as fh:
        fh.write(text)
    model = Darknet(path).eval()
    synth.fill_darknet_(model, tag)
    return model, text


def make_darknet(name, tag=None, trained_like=False):
    """Product-side Darknet module tree with deterministic weights (CPU tensors)."""
    from millieye_amd.yolov3.models import Darknet

    model = Darknet(cfg_path(name)).eval()
    synth.fill_darknet_(model, tag or name)
    if trained_like:
        synth.trained_like_(model, tag=(tag or name) + "/trained")
    return model


def frames(tag, n, s):
    return torch.from_numpy(synth.uniform(tag, (n, 3, s, s), 0.0, 1.0))


def max_rel_err(got, ref, floor=1.0):
    """max |got-ref| / max(|ref|, floor): with floor=1 this is allclose(rtol=tol, atol=tol); the
    1e-3 fp32 bar of BASELINE.json is applied to this number."""
    got, ref = got.double(), ref.double()
    return ((got - ref).abs() / ref.abs().clamp_min(floor)).max().item()


def assert_close(got, ref, tol, what):
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} != {tuple(ref.shape)}"
    bad = ~torch.isfinite(got)
    assert not bad.any(), f"{what}: {int(bad.sum())} non-finite values"
    err = max_rel_err(got, ref)
    assert err <= tol, f"{what}: max relative error {err:.3e} > {tol:.1e}"
    return err


def run_smoke(device="cuda:0"):
    """One tiny hot-path invocation (detector -> NMS -> fusion heads) checked against the oracle.
    Used by __graft_entry__.smoke(); tolerance 1e-3 like the parity tests."""
    from millieye_amd.my_models import Network, define_yolo
    from oracle import network_ref

    name, cfg, n, s, conf = "smoke", "yolov3-tiny-12", 2, 96, 0.2
    net = Network(define_yolo(cfg_path(cfg)), conf).eval()
    synth.fill_network_(net, name)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16)
    maps, rboxes = torch.from_numpy(maps), torch.from_numpy(rboxes)
    ref = network_ref.network_forward(cfg_text(cfg), sd, x, maps, rboxes, 0, conf_thresh=conf)
    net = net.to(device)
    with torch.no_grad():
        out = net(x.to(device), maps.to(device), rboxes.clone().to(device), 0)
        fm, yolo = net.base_detector(x.to(device))
    torch.cuda.synchronize()
    err = assert_close(out.cpu(), ref, 1e-3, "smoke Network.forward mode 0")
    return dict(rows=int(out.shape[0]), max_err=err, featuremap=tuple(fm.shape), yolo=tuple(yolo.shape))
