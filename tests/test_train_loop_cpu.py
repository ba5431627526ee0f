"""CPU (-m "not gpu"): the training *harness* (millieye_amd/train.py: stage-2 hand-over, freezing, step cadence,
checkpoint naming, per-epoch evaluate) against the REAL reference train.py run (golden), with the oracle doing
the arithmetic in place of the HIP library (the product Network has no CPU path)."""
import torch

from millieye_amd import cfgs
from millieye_amd.my_models import Network
from oracle import network_ref
from tests import train_loop_helpers as tl
from tests.golden.make_golden import LOOP_CASE


class _Inject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss, grads, *params):
        ctx.grads = grads
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        return (None, None) + tuple(None if g is None else g * grad_out for g in ctx.grads)


class OracleBackedNetwork(Network):
    """Same parameter container / attributes as the product, forward = oracle/network_ref.py on CPU."""

    def __init__(self, det, conf):
        super().__init__(det, conf)
        self.device = torch.device("cpu")
        self.cfg_text = cfgs.KNOWN[LOOP_CASE["cfg"]]()

    def forward(self, images, maps, radar_boxes, model_mode=0, targets=None):
        if torch.is_tensor(model_mode):
            model_mode, targets = 0, model_mode
        sd = self.state_dict()
        if targets is None:
            if model_mode == 2:
                self.refine_threshold_img = 1
            return network_ref.network_forward(self.cfg_text, sd, images, maps, radar_boxes, model_mode,
                                               self.conf_thresh, self.refine_threshold_img, self.refine_threshold_radar)
        res = network_ref.network_train_step(self.cfg_text, sd, images, maps, radar_boxes, targets,
                                             conf_thresh=self.conf_thresh)
        own = dict(self.named_buffers())
        with torch.no_grad():
            for k, v in res["buffers"].items():
                own[k].copy_(v)
            for k, v in own.items():
                if k.endswith("num_batches_tracked") and not k.startswith("base_detector."):
                    v += 1
        named = [(k, p) for k, p in self.named_parameters() if not k.startswith("base_detector.")]
        grads = [res["grads"].get(k) if p.requires_grad else None for k, p in named]
        loss = _Inject.apply(res["loss"], grads, *[p for _, p in named])
        metric = dict(true=res["n_pos"], total=len(res["masks"]))
        return loss, res["output"], metric, None


def test_train_loop_harness_matches_reference_script(tmp_path):
    net, frozen = tl.prepare(OracleBackedNetwork)
    hist = tl.run(net, tmp_path)
    tl.check(net, frozen, hist, tmp_path, loss_tol=1e-5, param_atol=2e-5, sum_tol=1e-6, ap_tol=1e-6)
