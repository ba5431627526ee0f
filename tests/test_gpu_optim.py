"""millieye_amd.optim.Adam / AdamW (one launch per step, csrc/optim.hip) against the classes the reference's loops build:
torch.optim.Adam(lr=5e-4) (module3_our_dataset/train.py:161) and torch.optim.AdamW(lr=1e-4) (module2/train.py:122)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(256, 980), (256,), (1, 256), (490, 256, 1, 1), (10, 4, 3, 3), (3,), (1,), (1025,), (2048,), (13, 7, 5)]


def _params(seed, shapes, device):
    g = np.random.RandomState(seed)
    return [torch.nn.Parameter(torch.from_numpy(g.standard_normal(s).astype(np.float32)).to(device)) for s in shapes]


def _grads(step, shapes, device, scale):
    g = np.random.RandomState(1000 + step)
    return [torch.from_numpy((scale * g.standard_normal(s)).astype(np.float32)).to(device) for s in shapes]


@pytest.mark.parametrize("kind,kw", [("Adam", dict(lr=5e-4)), ("AdamW", dict(lr=1e-4)), ("Adam", dict(lr=1e-3, weight_decay=0.05)),
                                     ("AdamW", dict(lr=2e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.1))])
def test_adam_step_vs_torch_single_tensor_rule(hip_lib, kind, kw):
    """30 steps on ten tensors (sizes around the 1024-element chunk, 1 element, 4-D weights), gradients spanning six orders of
    magnitude: parameters and both moments stay within a few fp32 roundings of torch's own implementation running the
    single-tensor rule on the CPU (foreach=False: the element-wise op sequence the kernel restates), and of its CUDA one."""
    from millieye_amd import optim
    mine_p = _params(3, SHAPES, "cuda")
    mine = getattr(optim, kind)(mine_p, **kw)
    ref_p = _params(3, SHAPES, "cpu")
    ref = getattr(torch.optim, kind)(ref_p, foreach=False, **kw)
    dev_p = _params(3, SHAPES, "cuda")
    dev = getattr(torch.optim, kind)(dev_p, **kw)
    for step in range(30):
        scale = 10.0 ** ((step % 6) - 4)
        for ps, where in ((mine_p, "cuda"), (ref_p, "cpu"), (dev_p, "cuda")):
            for p, g in zip(ps, _grads(step, SHAPES, where, scale)):
                p.grad = g
        mine.step(); ref.step(); dev.step()
    torch.cuda.synchronize()
    for i, (a, b, c) in enumerate(zip(mine_p, ref_p, dev_p)):
        scale_p = float(b.detach().abs().max())
        e_cpu, e_dev = float((a.detach().cpu() - b.detach()).abs().max()), float((a.detach() - c.detach()).abs().max())
        assert e_cpu <= 2e-6 * scale_p, (i, "parameter vs torch CPU", e_cpu, scale_p, float((c.detach().cpu() - b.detach()).abs().max()))
        assert e_dev <= 2e-6 * scale_p, (i, "parameter vs torch CUDA", e_dev, scale_p)
        for key in ("exp_avg", "exp_avg_sq"):
            ma, mb = mine.state[a][key].cpu(), ref.state[b][key]
            assert float((ma - mb).abs().max()) <= 1e-6 * float(mb.abs().max()) + 1e-30, (i, key)
        assert mine.state[a]["step"] == 30 and float(ref.state[b]["step"]) == 30.0


def test_adam_skips_parameters_without_gradient_and_keeps_their_step_counts(hip_lib):
    """A parameter with ``grad is None`` is left alone (torch: skipped, no decay, no step count) - the frozen detector under stage
    2's AdamW(model.parameters()); when it gets a gradient later its bias correction starts from its OWN step count."""
    from millieye_amd import optim
    shapes = [(300,), (5, 5), (2000,)]
    mine_p, ref_p = _params(5, shapes, "cuda"), _params(5, shapes, "cpu")
    mine, ref = optim.AdamW(mine_p, lr=1e-3), torch.optim.AdamW(ref_p, lr=1e-3, foreach=False)
    for step in range(8):
        for ps, where in ((mine_p, "cuda"), (ref_p, "cpu")):
            for j, (p, g) in enumerate(zip(ps, _grads(step, shapes, where, 1.0))):
                p.grad = None if (j == 1 and step < 4) else g
        mine.step(); ref.step()
        if step == 3:
            assert mine_p[1] not in mine.state or not mine.state[mine_p[1]]
            assert torch.equal(mine_p[1].detach().cpu(), _params(5, shapes, "cpu")[1].detach())
    assert [mine.state[p]["step"] for p in mine_p] == [8, 4, 8]
    for a, b in zip(mine_p, ref_p):
        assert float((a.detach().cpu() - b.detach()).abs().max()) <= 2e-6 * float(b.detach().abs().max())


def test_adam_checkpoints_are_interchangeable_with_torch(hip_lib):
    """state_dict() of this class loads into torch.optim.Adam and back; training continues on the same trajectory."""
    from millieye_amd import optim
    shapes = [(64, 32), (32,)]
    a_p, b_p = _params(9, shapes, "cuda"), _params(9, shapes, "cuda")
    mine, theirs = optim.Adam(a_p, lr=5e-4), torch.optim.Adam(b_p, lr=5e-4)
    for step in range(5):
        for ps in (a_p, b_p):
            for p, g in zip(ps, _grads(step, shapes, "cuda", 0.1)):
                p.grad = g
        mine.step(); theirs.step()
    sd = mine.state_dict()
    assert all(torch.is_tensor(st["step"]) and float(st["step"]) == 5.0 for st in sd["state"].values())
    assert all(st["step"] == 5 and not torch.is_tensor(st["step"]) for st in mine.state.values()), "state_dict() edited the live state"
    # mine -> torch, torch -> mine, then five more steps each: still the same trajectory
    c_p = [torch.nn.Parameter(p.detach().clone()) for p in a_p]
    into_torch = torch.optim.Adam(c_p, lr=5e-4)
    into_torch.load_state_dict(copy.deepcopy(sd))   # (load_state_dict keeps same-device tensors as they are: aliases otherwise)
    d_p = [torch.nn.Parameter(p.detach().clone()) for p in b_p]
    into_mine = optim.Adam(d_p, lr=5e-4)
    into_mine.load_state_dict(copy.deepcopy(theirs.state_dict()))
    assert all(st["step"] == 5 for st in into_mine.state.values())
    for step in range(5, 10):
        for ps in (a_p, c_p, d_p):
            for p, g in zip(ps, _grads(step, shapes, "cuda", 0.1)):
                p.grad = g
        mine.step(); into_torch.step(); into_mine.step()
    for a, c, d in zip(a_p, c_p, d_p):
        tol = 2e-6 * float(a.detach().abs().max())
        assert float((a.detach() - c.detach()).abs().max()) <= tol and float((a.detach() - d.detach()).abs().max()) <= tol


def test_adam_refuses_what_it_does_not_implement(hip_lib):
    from millieye_amd import hip, optim
    p = _params(1, [(4,)], "cuda")
    with pytest.raises(NotImplementedError):
        optim.Adam(p, amsgrad=True)
    with pytest.raises(NotImplementedError):
        optim.Adam(p, maximize=True)
    with pytest.raises(ValueError):
        optim.Adam(p, betas=(0.4, 0.999))
    cpu = optim.Adam(_params(1, [(4,)], "cpu"))
    cpu.param_groups[0]["params"][0].grad = torch.ones(4)
    with pytest.raises(hip.MeError):
        cpu.step()   # no CPU path
