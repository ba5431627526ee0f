"""``optimizer.step()`` of the reference's training loops as one HIP launch.

The reference builds ``torch.optim.Adam(model.parameters(), lr=5e-4)`` for stage 3 (module3_our_dataset/train.py:161, stepped at
:196-197) and ``torch.optim.AdamW(model.parameters(), lr=1e-4)`` for stage 2 (module2/train.py:122, :150-151).  ``Adam`` / ``AdamW``
below take the same constructor arguments, keep the same per-parameter state (``step``, ``exp_avg``, ``exp_avg_sq``) and do the
arithmetic of ``torch/optim/adam.py:_single_tensor_adam`` element by element in fp32 - but for every parameter tensor of a group in
ONE launch of ``me_adam_step_f32`` (csrc/optim.hip; 64 tensors per launch).  The stage-3 state is 40 small tensors (100 153
parameters): torch's fused implementation spent 96 us of GPU time per step on it, this one a few.

``state_dict()`` / ``load_state_dict()`` exchange checkpoints with the torch classes (``step`` is written as a 0-dim fp32 tensor
like torch does, and read back from either a tensor or a number).  Parameters must be fp32 CUDA tensors: there is no CPU path -
the loops (millieye_amd/train.py, module2/train.py) fall back to the torch classes themselves when a model is not on the GPU
(the host-logic tests).  ``amsgrad``, ``maximize``, ``capturable`` and ``differentiable`` are not implemented and refused.
"""
import ctypes as C

import torch

from . import hip


class _AdamBase(torch.optim.Optimizer):
    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **unsupported):
        for key in ("maximize", "capturable", "differentiable"):
            if unsupported.pop(key, False):
                raise NotImplementedError(f"millieye_amd.optim: {key}=True is not implemented")
        for key in ("foreach", "fused"):   # implementation hints of the torch classes: this class IS the fused implementation
            unsupported.pop(key, None)
        if unsupported:
            raise TypeError(f"unexpected arguments {sorted(unsupported)}")
        if amsgrad:
            raise NotImplementedError("millieye_amd.optim: amsgrad=True is not implemented")
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError("millieye_amd.optim: a tensor learning rate is not implemented")
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.5 < betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]} (this implementation needs 0.5 < beta1 < 1)")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._desc = hip.AdamDesc()
        self._chunk = None

    # ---- checkpoints in the torch classes' format ------------------------------------------------------------------------
    def state_dict(self):
        sd = super().state_dict()
        state = {}
        for key, st in sd["state"].items():   # (the per-parameter dicts in there are the live ones: copies, not edits)
            st = dict(st)
            if "step" in st and not torch.is_tensor(st["step"]):
                st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
            state[key] = st
        sd["state"] = state
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if "step" in st:
                st["step"] = int(float(st["step"]))

    # ---- the step --------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = hip.lib()
        if self._chunk is None:
            self._chunk = int(lib.me_adam_chunk())
        chunk = self._chunk
        d = self._desc
        state = self.state
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr = float(group["lr"])
            # the parameters of one launch share the step count (they do unless some had no gradient in an earlier step)
            by_step = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                st = state[p]
                if not st:
                    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                        raise hip.MeError("millieye_amd.optim: parameters must be contiguous fp32 CUDA tensors (no CPU path; "
                                          "use torch.optim for a model that is not on the GPU)")
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                    g = g.to(device=p.device, dtype=torch.float32).contiguous()
                st["step"] += 1
                by_step.setdefault(st["step"], []).append((p, g, st))
            d.decoupled = 1 if self._decoupled else 0
            # the scalars torch's Python code hands to the tensor ops, computed in double here as there
            wd = float(group["weight_decay"])
            d.beta2, d.eps, d.weight_decay, d.decay = beta2, group["eps"], wd, 1.0 - lr * wd
            d.one_minus_beta1, d.one_minus_beta2 = 1.0 - beta1, 1.0 - beta2
            for t, items in by_step.items():
                d.neg_step_size = -(lr / (1.0 - beta1 ** t))
                d.bias_correction2_sqrt = (1.0 - beta2 ** t) ** 0.5
                for lo in range(0, len(items), hip.ADAM_MAX_TENSORS):
                    part = items[lo:lo + hip.ADAM_MAX_TENSORS]
                    chunks = 0
                    for i, (p, g, st) in enumerate(part):
                        n = p.numel()
                        d.param[i] = p.data_ptr()
                        d.grad[i] = g.data_ptr()
                        d.exp_avg[i] = st["exp_avg"].data_ptr()
                        d.exp_avg_sq[i] = st["exp_avg_sq"].data_ptr()
                        d.numel[i] = n
                        d.first_chunk[i] = chunks
                        chunks += (n + chunk - 1) // chunk
                    d.count = len(part)
                    with torch.cuda.device(part[0][0].device):
                        hip.check(lib.me_adam_step_f32(C.byref(d), hip.stream_ptr()), "me_adam_step_f32")
        return loss


class Adam(_AdamBase):
    """``torch.optim.Adam`` (reference module3_our_dataset/train.py:161): L2 weight decay is added to the gradient."""
    _decoupled = False


class AdamW(_AdamBase):
    """``torch.optim.AdamW`` (reference module2/train.py:122): decoupled weight decay, default 1e-2 like the torch class."""
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
