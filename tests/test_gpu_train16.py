"""-m gpu: the detector training step in the 16-bit storage modes (millieye_amd/detector_train16.py, csrc/train_h16.hip):
16-bit activations and activation gradients, fp32 master weights / weight gradients / per-channel sums.  Bars: the
activation-gradient kernel against the fp32 kernel on the same 16-bit inputs (one storage ulp; sums 1e-4), and the whole step
against the fp32 HIP step - which the other suites pin to the oracle and the reference - loss within 1 %, every parameter
gradient at cosine >= 0.99."""
import pytest
import torch

from millieye_amd import synth
from tests import parity_helpers as ph

pytestmark = pytest.mark.gpu
HALVES = {"bf16": torch.bfloat16, "f16": torch.float16}
ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}   # half a storage step, relative


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("rows,c,act,bn", [(2 * 52 * 52, 128, 1, True), (3 * 13 * 13, 1024, 1, True), (1000, 64, 0, False), (77, 32, 1, True)])
def test_affine_act_bwd_h16_vs_f32_kernel(hip_lib, half, rows, c, act, bn):
    from millieye_amd import hip
    dt = HALVES[half]
    dev = torch.device("cuda")
    lib = hip.lib()
    y16 = torch.from_numpy(synth.uniform(f"a16/y{rows}", (rows, c), -2, 2)).to(dev).to(dt)
    g16 = torch.from_numpy(synth.uniform(f"a16/g{rows}", (rows, c), -1, 1)).to(dev).to(dt)
    scale = torch.from_numpy(synth.uniform("a16/s", (c,), 0.5, 1.5)).to(dev)
    gam = torch.from_numpy(synth.uniform("a16/ga", (c,), 0.5, 1.5)).to(dev) if bn else None
    bet = torch.from_numpy(synth.uniform("a16/be", (c,), -0.5, 0.5)).to(dev) if bn else None
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    # fp32 kernel on the float copies of the SAME 16-bit values
    y32, g32 = y16.float(), g16.float()
    dc32, ds32 = torch.empty_like(y32), torch.empty(c, device=dev)
    dg32 = torch.empty(c, device=dev) if bn else None
    ws = torch.empty(lib.me_affine_bwd_workspace_bytes(rows, c), dtype=torch.uint8, device=dev)
    hip.check(lib.me_affine_act_bwd_f32(y32.data_ptr(), c, g32.data_ptr(), c, rows, c, scale.data_ptr() if bn else None, p(gam), p(bet),
                                        act, dc32.data_ptr(), c, ds32.data_ptr(), p(dg32), ws.data_ptr(), hip.stream_ptr()), "f32")
    dc16 = torch.full((rows, c), 7.0, device=dev, dtype=dt)
    ds16 = torch.empty(c, device=dev)
    dg16 = torch.empty(c, device=dev) if bn else None
    ws16 = torch.empty(lib.me_affine_bwd_h16_workspace_bytes(rows, c), dtype=torch.uint8, device=dev)
    hip.check(lib.me_affine_act_bwd_h16(y16.data_ptr(), c, g16.data_ptr(), c, rows, c, scale.data_ptr() if bn else None, p(gam), p(bet),
                                        act, dc16.data_ptr(), c, ds16.data_ptr(), p(dg16), ws16.data_ptr(), hip.HALF_TYPES[dt],
                                        hip.stream_ptr()), "h16")
    torch.cuda.synchronize()
    # dc: the fp32 result rounded once to the storage type
    assert torch.equal(dc16, dc32.to(dt)), f"dc differs from the rounded fp32 result on {int((dc16 != dc32.to(dt)).sum())} elements"
    torch.testing.assert_close(ds16, ds32, rtol=1e-4, atol=1e-4 * float(ds32.abs().max()))
    if bn:
        torch.testing.assert_close(dg16, dg32, rtol=1e-4, atol=1e-4 * float(dg32.abs().max()))
    # in place (dc aliases dy): same result
    g_inplace = g16.clone()
    hip.check(lib.me_affine_act_bwd_h16(y16.data_ptr(), c, g_inplace.data_ptr(), c, rows, c, scale.data_ptr() if bn else None, p(gam),
                                        p(bet), act, g_inplace.data_ptr(), c, ds16.data_ptr(), p(dg16), ws16.data_ptr(),
                                        hip.HALF_TYPES[dt], hip.stream_ptr()), "h16 in place")
    assert torch.equal(g_inplace, dc16)
    # the two halves apart (the detector backward runs the second one on its side stream): dc first, the sums later - the same bits
    dc_t, ds_t = torch.full((rows, c), 7.0, device=dev, dtype=dt), torch.full((c,), 3.0, device=dev)
    dg_t = torch.full((c,), 3.0, device=dev) if bn else None
    hip.check(lib.me_affine_act_bwd_h16(y16.data_ptr(), c, g16.data_ptr(), c, rows, c, scale.data_ptr() if bn else None, p(gam), p(bet),
                                        act, dc_t.data_ptr(), c, None, None, ws16.data_ptr(), hip.HALF_TYPES[dt], hip.stream_ptr()), "h16 dc only")
    hip.check(lib.me_affine_bwd_h16_sums(ws16.data_ptr(), rows, c, ds_t.data_ptr(), p(dg_t), hip.stream_ptr()), "h16 sums")
    assert torch.equal(dc_t, dc16) and torch.equal(ds_t, ds16) and (not bn or torch.equal(dg_t, dg16))
    dc_f, ds_f = torch.full((rows, c), 7.0, device=dev), torch.full((c,), 3.0, device=dev)
    dg_f = torch.full((c,), 3.0, device=dev) if bn else None
    hip.check(lib.me_affine_act_bwd_f32(y32.data_ptr(), c, g32.data_ptr(), c, rows, c, scale.data_ptr() if bn else None, p(gam), p(bet),
                                        act, dc_f.data_ptr(), c, None, None, ws.data_ptr(), hip.stream_ptr()), "f32 dc only")
    hip.check(lib.me_affine_bwd_sums_f32(ws.data_ptr(), rows, c, int(c % 4 == 0), ds_f.data_ptr(), p(dg_f), hip.stream_ptr()), "f32 sums")
    assert torch.equal(dc_f, dc32) and torch.equal(ds_f, ds32) and (not bn or torch.equal(dg_f, dg32))
    with pytest.raises(hip.MeError):   # channels % 8
        lib_rc = lib.me_affine_act_bwd_h16(y16.data_ptr(), c, g16.data_ptr(), c, rows, c - 4, None, None, None, act, dc16.data_ptr(), c,
                                           ds16.data_ptr(), None, ws16.data_ptr(), hip.HALF_TYPES[dt], hip.stream_ptr())
        hip.check(lib_rc, "bad channels")


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("n,h,cin,cout,k,s", [(2, 26, 64, 128, 3, 1), (2, 26, 128, 256, 3, 2), (3, 13, 256, 128, 1, 1), (1, 20, 32, 64, 3, 1),
                                              (2, 13, 512, 256, 1, 1), (1, 52, 128, 128, 3, 1)])
def test_conv_wgrad_h16_vs_fp32_kernel(hip_lib, half, n, h, cin, cout, k, s):
    """``me_conv_wgrad_h16`` (16-bit operands, 16-bit MFMA, fp32 accumulation) against the fp32 weight-gradient kernels on float
    copies of the SAME 16-bit values: the products are exact in fp32 either way, only the summation order differs (1e-4 of the
    tensor's maximum); both the 64 x 64 and the 128 x 128 tile, every tap, stride 2, ragged pixel counts."""
    from millieye_amd import hip
    dt = HALVES[half]
    pad = (k - 1) // 2
    ho = (h + 2 * pad - k) // s + 1
    x = torch.from_numpy(synth.uniform(f"w16/x{h}{cin}", (n, h, h, cin), -1, 1)).cuda().to(dt)
    dy = torch.from_numpy(synth.uniform(f"w16/d{h}{cout}", (n, ho, ho, cout), -1, 1)).cuda().to(dt)
    ref = hip.conv_wgrad(x.float(), dy.float(), k, s, pad, oihw=True)
    got = hip.conv_wgrad_h16(x, dy, k, s, pad, oihw=True)
    assert got.dtype == torch.float32 and got.shape == ref.shape == (cout, cin, k, k)
    err = float((got - ref).abs().max())
    assert err <= 1e-4 * float(ref.abs().max()), f"max err {err:.3e} vs max {float(ref.abs().max()):.3e}"
    got2 = hip.conv_wgrad_h16(x, dy, k, s, pad, oihw=True)
    assert torch.equal(got, got2)   # fixed-order sums: deterministic
    ohwi = hip.conv_wgrad_h16(x, dy, k, s, pad, oihw=False)
    assert torch.equal(ohwi.permute(0, 3, 1, 2).contiguous(), got)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cfg,n,s", [("yolov3", 2, 128)])
def test_detector_step_16bit_vs_fp32_hip_step(hip_lib, dtype, cfg, n, s):
    """The mixed-precision step against the fp32 HIP step on the same inputs: Darknet-53 at 128 px (stride-2 parity data
    gradients, 23 shortcuts, three scales, upsample + route concats; at 64 px the deepest maps are 2 x 2 positions and a weight
    gradient is the sum of eight products - measured cosine 0.988 on one 13-stage 1x1 filter there)."""
    name = f"t16/{cfg}"
    targets = torch.tensor([[0, 0, 0.30, 0.40, 0.20, 0.30], [1, 0, 0.70, 0.60, 0.50, 0.40], [1, 0, 0.52, 0.48, 0.10, 0.15]])
    if cfg == "yolov3":
        targets[:, 1] = torch.tensor([3.0, 17.0, 60.0])
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s))).cuda()
    ref = ph.make_darknet(cfg, tag=name).cuda().eval()
    loss32, fm32, yo32 = ref(x, targets)
    loss32.backward()
    model = ph.make_darknet(cfg, tag=name).cuda().eval()
    model.compute_dtype = dtype
    loss16, fm16, yo16 = model(x, targets)
    assert loss16.requires_grad and fm16.shape == fm32.shape and yo16.shape == yo32.shape and fm16.dtype == torch.float32
    rel = abs(float(loss16.detach()) - float(loss32.detach())) / abs(float(loss32.detach()))
    (2.0 * loss16).backward()   # an upstream gradient of 2 must scale every gradient
    # (f16: measured worst cosine 0.9987 - 0.9993 from run to run on the 4 x 4 maps of this size - the layer tiles are re-tuned per
    #  process and another K split rounds a few 16-bit activations the other way; at the bench shape every tensor is >= 0.99995)
    tol_loss, tol_cos = (0.01, 0.99) if dtype == "bf16" else (0.002, 0.998)
    assert rel <= tol_loss, f"loss {float(loss16):.5f} vs fp32 {float(loss32):.5f}: {rel:.3%}"
    worst = (1.0, None)
    seen = 0
    bad = []
    for (k, p16), (_k, p32) in zip(model.named_parameters(), ref.named_parameters()):
        assert p16.grad is not None and p16.grad.dtype == torch.float32 and p16.grad.shape == p32.grad.shape, k
        assert bool(torch.isfinite(p16.grad).all()), k
        g16, g32 = p16.grad / 2.0, p32.grad
        if float(g32.norm()) < 1e-12:
            continue
        c = _cos(g16, g32)
        nr = float(g16.norm() / g32.norm())
        if c < worst[0]:
            worst = (c, k)
        if not (c >= tol_cos and 0.9 <= nr <= 1.1):
            bad.append(f"{k}: cosine {c:.4f}, norm ratio {nr:.3f}")
        seen += 1
    print(f"[{cfg} {dtype}] loss rel {rel:.2e}; worst gradient cosine {worst[0]:.5f} ({worst[1]}); {seen} tensors")
    assert not bad, "; ".join(bad[:12]) + f" ({len(bad)} of {seen})"
    assert seen >= 30
    # deterministic
    model.zero_grad()
    l1, _, _ = model(x, targets)
    l1.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    l2, _, _ = model(x, targets)
    l2.backward()
    assert float(l1.detach()) == float(l2.detach()) and all(torch.equal(g1[k], p.grad) for k, p in model.named_parameters())
    # an optimizer step on the fp32 master weights lowers the (16-bit forward's) loss; the next forward sees the new weights
    torch.optim.SGD(model.parameters(), lr=1e-4).step()
    model.zero_grad()
    loss_b, _, _ = model(x, targets)
    assert float(loss_b.detach()) < float(l2.detach())


def test_16bit_training_refuses_what_it_cannot_do(hip_lib):
    """The tiny cfgs (16-channel stem, max-pooling) stay float32 training paths: refused loudly, not computed wrongly."""
    model = ph.make_darknet("yolov3-tiny-12", tag="t16/tiny").cuda().eval()
    model.compute_dtype = "bf16"
    x = torch.from_numpy(synth.uniform("t16/tiny/x", (2, 3, 96, 96))).cuda()
    with pytest.raises(NotImplementedError):
        model(x, torch.tensor([[0, 0, 0.5, 0.5, 0.2, 0.2]]))
    model.compute_dtype = "f32"
    loss, _, _ = model(x, torch.tensor([[0, 0, 0.5, 0.5, 0.2, 0.2]]))   # (the float32 path still trains it)
    loss.backward()
    assert model.module_list[0][0].weight.grad is not None


def _step_errors(named_params, scale, ref32, rst):
    """Per parameter tensor: the HIP gradient's relative L2 distance from the fp32 oracle, the restatement's distance from the
    fp32 oracle, and the cosine of the HIP gradient with the restatement's."""
    rows = []
    for k, p in named_params:
        g = (p.grad / scale).cpu().double()
        a, r = ref32[k].double(), rst[k].double()
        if float(a.norm()) < 1e-12:
            continue
        rows.append((k, float((g - a).norm() / a.norm()), float((r - a).norm() / a.norm()), _cos(g, r), _cos(g, a)))
    return rows


@pytest.mark.parametrize("dtype,n,s", [("bf16", 2, 128), ("f16", 2, 128), ("bf16", 8, 416)])
def test_detector_step_16bit_vs_oracle_restatement(hip_lib, dtype, n, s):
    """The 16-bit training step against an INDEPENDENT check (until round 6 it was compared with the fp32 HIP
    step only): ``oracle.darknet_ref.darknet_train_step(storage=...)`` = fp32 CPU autograd through a forward rounded at this
    path's rounding points, with the activation gradients rounded where ``detector_train16.py`` stores them, and the plain fp32
    oracle step (pinned to the reference's autograd by ``yololoss_*.npz``).  Darknet-53, trained-like weights, 128 px batch 2 in
    both formats and the bench's shape (416 px, batch 8) in bf16.  Bars:
      * the storage format, not the kernels, sets the error: the mean over the 222 parameter tensors of the HIP gradient's
        relative distance from the fp32 oracle is within 1.25 x + 1e-4 of the restatement's own distance (the bar of
        ``test_detector_bf16``), and so is the loss;
      * HIP and restatement agree tensor by tensor: at the bench's shape every cosine >= 0.998 (measured worst 0.99846: the 1x1
        convolution in front of the second upsample), median >= 0.9999 (measured 0.99997); at 128 px the deepest maps are 4 x 4
        positions x 2 frames and a rounding flip of one activation moves a whole filter's gradient, so the per-tensor bar there is
        0.98 (bf16; measured worst 0.9886 on a 512-channel BatchNorm weight) / 0.997 (f16) and the median is >= 0.995 (bf16, measured 0.9972 - 0.9988 from run to run: the layer tiles are tuned per process) / 0.999 (f16, 0.9997 - 0.9999)."""
    import os
    from millieye_amd import cfgs
    from oracle import darknet_ref
    name = f"t16r/{n}/{s}"
    cpu_model = ph.make_darknet("yolov3", tag=name, trained_like=True)
    targets = torch.tensor([[i % n, (3 * i) % 80, 0.3 + 0.04 * (i % 8), 0.4 + 0.03 * (i % 5), 0.2, 0.3] for i in range(max(n, 3))],
                           dtype=torch.float32)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    torch.set_num_threads(min(64, len(os.sched_getaffinity(0))))
    text, sd = cfgs.KNOWN["yolov3"](), cpu_model.state_dict()
    loss32, g32 = darknet_ref.darknet_train_step(text, sd, x, targets)
    loss_r, g_r = darknet_ref.darknet_train_step(text, sd, x, targets, storage=dtype)
    model = ph.make_darknet("yolov3", tag=name, trained_like=True).cuda().eval()
    model.compute_dtype = dtype
    loss, _fm, _yo = model(x.cuda(), targets)
    loss.backward()
    torch.cuda.synchronize()
    rel_h = abs(float(loss.detach()) - float(loss32)) / abs(float(loss32))
    rel_r = abs(float(loss_r) - float(loss32)) / abs(float(loss32))
    rows = _step_errors(list(model.named_parameters()), 1.0, g32, g_r)
    assert len(rows) == 222
    e_h = sum(r[1] for r in rows) / len(rows)
    e_r = sum(r[2] for r in rows) / len(rows)
    cos = sorted(r[3] for r in rows)
    worst = min(rows, key=lambda r: r[3])
    print(f"[{dtype} {n}x{s}] loss: HIP {rel_h:.2e} / restatement {rel_r:.2e} from fp32; gradients: mean distance from fp32 HIP {e_h:.3e} / "
          f"restatement {e_r:.3e}; cosine HIP vs restatement: median {cos[len(cos) // 2]:.5f}, worst {worst[3]:.5f} ({worst[0]})")
    assert rel_h <= 1.25 * rel_r + (1e-3 if dtype == "f16" else 5e-3), (rel_h, rel_r)
    assert e_h <= 1.25 * e_r + 1e-4, f"HIP gradients {e_h:.3e} from the fp32 oracle, the restatement {e_r:.3e}"
    per_tensor = 0.998 if s >= 416 else (0.98 if dtype == "bf16" else 0.997)
    bad = [f"{r[0]}: {r[3]:.4f}" for r in rows if r[3] < per_tensor]
    assert not bad, "; ".join(bad[:10]) + f" ({len(bad)} of {len(rows)} below {per_tensor})"
    assert cos[len(cos) // 2] >= (0.9999 if s >= 416 else 0.995 if dtype == "bf16" else 0.999)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cfg", ["mini32", "yolov3"])
def test_16bit_training_with_train_mode_batchnorm(hip_lib, dtype, cfg):
    """``model.train()`` in a 16-bit storage mode (refused until round 6): float32 batch statistics over the raw float32 sums of
    the 16-bit convolution, normalisation / its gradient in float32 (``me_bn_train_fwd_f32`` / ``_bwd_f32``), activations and
    activation gradients stored in the 16-bit type - against the oracle's restatement of exactly that
    (``darknet_train_step(training=True, storage=...)``) and against the fp32 oracle step that
    ``yololoss_tiny12_s96_n2_bntrain.npz`` pins to the reference's own ``model.train()`` run: loss, every gradient, every running
    statistic (momentum 0.9), ``num_batches_tracked``, and the no-autograd form (``Darknet.forward(x)`` under ``model.train()``).

    Batch-statistics BatchNorm subtracts the mean and the xhat-component of every activation gradient; the YOLO loss's gradient is
    mostly such a common component (``noobj_scale`` 100 over every cell), so what is left carries the 16-bit rounding noise many
    times amplified: against fp32 the RESTATEMENT's own gradients sit at cosine ~0.97 (bf16) on a 20-module cfg and lose all
    resemblance over Darknet-53's 72 normalisations (measured mean distance 0.74 bf16 / 0.30 f16, HIP and restatement alike).
    So: tensor-by-tensor agreement of HIP with the restatement on the shallow ``mini32`` cfg (tests/parity_helpers.py: every
    block kind of yolov3.cfg, 32-multiple channels); on Darknet-53 the format-sets-the-error bar, the loss and the statistics."""
    import os
    from millieye_amd import cfgs
    from oracle import darknet_ref
    mini = cfg == "mini32"
    name, n, s = f"t16bn/{cfg}", 4, (64 if mini else 128)
    if mini:
        cpu_model, text = ph.make_mini32(name, size=s)
        targets = torch.tensor([[0, 1, 0.30, 0.40, 0.20, 0.30], [1, 2, 0.70, 0.60, 0.50, 0.40], [3, 0, 0.52, 0.48, 0.10, 0.15]])
    else:
        cpu_model, text = ph.make_darknet("yolov3", tag=name), cfgs.KNOWN["yolov3"]()
        targets = torch.tensor([[0, 3, 0.30, 0.40, 0.20, 0.30], [1, 17, 0.70, 0.60, 0.50, 0.40], [3, 60, 0.52, 0.48, 0.10, 0.15]])
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    torch.set_num_threads(min(64, len(os.sched_getaffinity(0))))
    sd = cpu_model.state_dict()
    loss32, g32, b32 = darknet_ref.darknet_train_step(text, sd, x, targets, training=True)
    loss_r, g_r, b_r = darknet_ref.darknet_train_step(text, sd, x, targets, training=True, storage=dtype)
    model = (ph.make_mini32(name, size=s)[0] if mini else ph.make_darknet("yolov3", tag=name)).cuda().train()
    model.compute_dtype = dtype
    loss, fm, yo = model(x.cuda(), targets)
    assert loss.requires_grad and fm.dtype == torch.float32 and fm.shape[0] == n
    loss.backward()
    torch.cuda.synchronize()
    rel_h = abs(float(loss.detach()) - float(loss32)) / abs(float(loss32))
    rel_r = abs(float(loss_r) - float(loss32)) / abs(float(loss32))
    rows = _step_errors(list(model.named_parameters()), 1.0, g32, g_r)
    e_h = sum(r[1] for r in rows) / len(rows)
    e_r = sum(r[2] for r in rows) / len(rows)
    cos = sorted(r[3] for r in rows)
    worst = min(rows, key=lambda r: r[3])
    print(f"[bn-train {cfg} {dtype}] loss: HIP {rel_h:.2e} / restatement {rel_r:.2e} from fp32; gradients: HIP {e_h:.3e} / restatement {e_r:.3e} "
          f"from fp32; cosine HIP vs restatement: median {cos[len(cos) // 2]:.5f}, worst {worst[3]:.5f} ({worst[0]})")
    assert rel_h <= 1.25 * rel_r + (2e-3 if dtype == "f16" else 1e-2), (rel_h, rel_r)
    assert e_h <= 1.25 * e_r + 1e-3, f"HIP gradients {e_h:.3e} from the fp32 oracle, the restatement {e_r:.3e}"
    if mini:
        bar = 0.97 if dtype == "bf16" else 0.9995   # measured worst 0.979 / 0.99994, medians 0.992 / 0.99998
        bad = [f"{r[0]}: {r[3]:.4f}" for r in rows if r[3] < bar]
        assert not bad, "; ".join(bad[:10]) + f" ({len(bad)} of {len(rows)} below {bar})"
    got = model.state_dict()
    for k, v in b_r.items():
        ref = v.double()
        err = float((got[k].cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
        assert err <= (2e-2 if dtype == "bf16" else 3e-3) * (1 if mini else 5), f"{k}: running statistic {err:.3e} from the restatement"   # (Darknet-53, measured worst 4.6e-2 / 6.3e-3: the variance of a 64-sample channel 78 layers deep)
        err32 = float((got[k].cpu().double() - b32[k].double()).abs().max() / b32[k].double().abs().max().clamp_min(1e-6))
        assert err32 <= (5e-2 if dtype == "bf16" else 1e-2) * (1 if mini else 4), f"{k}: running statistic {err32:.3e} from the fp32 oracle"
    assert all(int(v) == 1 for k, v in got.items() if k.endswith("num_batches_tracked"))
    # no autograd: the batch-statistics forward of the same storage mode (running statistics move again)
    before = {k: v.clone() for k, v in got.items() if "running_mean" in k}
    with torch.no_grad():
        fm2, yo2 = model(x.cuda())
    assert yo2.shape == yo.shape and bool(torch.isfinite(yo2).all())
    assert any(not torch.equal(model.state_dict()[k], v) for k, v in before.items())


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_direct_16bit_weight_pack_equals_the_fp32_pack_rounded(hip_lib, dtype):
    """me_pack_conv_batch_f32 with 16-bit outputs (me_pack_desc.ohwi16 / rot16 / parity16, ABI 11) against the fp32 pack + one RNE
    conversion: every OHWI / rotated / parity tensor of Darknet-53 equal, the folded scale / shift and the fp32 copies the step
    reads (stem, detection convolutions) equal to the fp32 pack's."""
    from millieye_amd.detector_train16 import _Weights16
    half = HALVES[dtype]
    model = ph.make_darknet("yolov3", tag="t16/pack", trained_like=True).cuda().eval()
    eng, defs, dev = model.engine, model.module_defs, torch.device("cuda", torch.cuda.current_device())
    two = _Weights16()
    eng.refresh_train_weights(dev)
    two.refresh(eng, defs, half, dev)
    torch.cuda.synchronize()
    want = {k: v.clone() for k, v in two.dst.items()}
    convs = [i for i, d in enumerate(defs) if d["type"] == "convolutional"]
    f32 = {i: (eng._conv_weights(i).wgt.clone(), eng._conv_weights(i).scale.clone(), eng._conv_weights(i).shift.clone()) for i in convs}
    for i in convs:   # poison what the direct pack must write
        cw = eng._conv_weights(i)
        cw.scale.fill_(7.0)
        cw.shift.fill_(7.0)
        if cw.wgt.shape[3] <= 4 or (i + 1 < len(defs) and defs[i + 1]["type"] == "yolo"):
            cw.wgt.fill_(7.0)
    one = _Weights16()
    assert one.pack_direct(eng, defs, half, dev) and one.direct
    torch.cuda.synchronize()
    assert sorted(one.dst) == sorted(want) and len(want) >= 140
    for k, v in want.items():
        assert torch.equal(one.dst[k], v), k
    for i in convs:
        cw = eng._conv_weights(i)
        assert torch.equal(cw.scale, f32[i][1]) and torch.equal(cw.shift, f32[i][2]), i
        if cw.wgt.shape[3] <= 4 or (i + 1 < len(defs) and defs[i + 1]["type"] == "yolo"):
            assert torch.equal(cw.wgt, f32[i][0]), i
        assert cw._stamp is None   # (the engine's fp32 copies are packed again when something asks for them)
    # ... and something does: the fp32 inference path after the 16-bit pack sees current weights
    x = torch.from_numpy(synth.uniform("t16/pack/x", (1, 3, 96, 96))).cuda()
    with torch.no_grad():
        a = model(x)
    ref = ph.make_darknet("yolov3", tag="t16/pack", trained_like=True).cuda().eval()
    with torch.no_grad():
        b = ref(x)
    a_rows, b_rows = (a[1] if isinstance(a, tuple) else a), (b[1] if isinstance(b, tuple) else b)
    assert torch.equal(a_rows, b_rows)
