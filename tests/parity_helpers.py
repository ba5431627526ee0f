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
