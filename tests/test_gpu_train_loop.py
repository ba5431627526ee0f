"""GPU: the whole stage-3 training harness on the HIP path (Network.forward with targets, loss.backward(), Adam,
checkpoints, per-epoch evaluate) against the REAL reference train.py run recorded in
tests/golden/trainloop_tiny12_s160.npz.

Tolerances: losses 1e-3 relative (north_star's fp32 bound).  Parameters after three Adam steps: Adam's update is
``lr * m / (sqrt(v) + eps)`` = ``+-lr`` per step for every element, whatever the gradient's magnitude, so an element
whose gradient is at rounding-noise level may step the other way: ``|diff| <= 2 * lr * steps = 3e-3`` is the bound
for such elements, the *mean* over a tensor (``step_sums``) stays within 2e-4 per element.  precision / recall / f1 /
box counts of every ``evaluate`` and the AP of the first one must agree to 2e-3; AP is a function of the *rank order*
of the scores, and after the third step neighbouring detections whose scores differ by 2e-4 (0.91806 / 0.91784 in the
golden rows) may swap, which moves this 29-detection AP by one PR-curve corner (0.006): later APs get 2e-2."""
import pytest
import torch

from tests import train_loop_helpers as tl

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("optimizer", ["torch", "hip"])
def test_train_loop_on_gpu_matches_reference_script(hip_lib, tmp_path, optimizer):
    """optimizer = "hip": millieye_amd.optim.Adam (one launch per step; the loop's default on the GPU) instead of torch.optim.Adam -
    the same trajectory of the REAL reference's train.py run (parameter sums after every step, losses, final weights)."""
    from millieye_amd.my_models import Network
    from millieye_amd import optim
    net, frozen = tl.prepare(Network)
    net = net.to(net.device)
    assert net.device.type == "cuda"
    hist = tl.run(net, tmp_path, optimizer_cls=optim.Adam if optimizer == "hip" else None)
    tl.check(net, frozen, hist, tmp_path, loss_tol=1e-3, param_atol=3.2e-3, sum_tol=2e-4, ap_tol=2e-3, late_ap_tol=2e-2)
    # every trainable head tensor moved, frozen ones did not
    g = tl.golden()
    sd = net.state_dict()
    for k in frozen:
        assert torch.equal(sd[k].cpu(), torch.from_numpy(g["final/" + k])), k
