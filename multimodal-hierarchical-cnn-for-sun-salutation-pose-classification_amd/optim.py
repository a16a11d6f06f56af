"""Adam for the plan-backed models: optimizer step and operand re-packing in one pass.

Same update rule and constructor arguments as the reference's optimizer
(`torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)`,
/root/reference/Quadtree_from scratch/Quadtree_train.py:45; resnet/train_cnn_model.py:65): L2 weight decay
added to the gradient, bias-corrected moments, no amsgrad / maximize.  The arithmetic runs in
csrc/pack.hip: the conv / linear weights (99.6 % of the parameters) are updated inside the
one-launch kernel that re-packs them into the MFMA operand layouts, so masters, moments and packed
copies are each read / written once per step; the remaining small tensors take one multi-tensor
launch.  SURVEY.md 8(f) rank 1.

    opt = FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-4, model=model)

Without `model=` (or for parameters the plan does not know) it is a plain fused multi-tensor Adam.
"""
import ctypes

import torch

from . import _lib
from .engine import AdamDesc, AdamItem


# Parameter updates made through raw pointers (qt_adam_multi) do not bump torch's version counters: consumers that cache
# packed copies of parameters (video3d._ConvBlock.pack) compare this counter as well.
_raw_updates = 0


def raw_update_count():
    return _raw_updates


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, model=None):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("FusedAdam: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._model = model

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        engine = getattr(self._model, "_engine", None) if self._model is not None else None
        plan_index = {}
        if engine is not None and self._model._param_list is not None:
            plan_index = {id(p): i for p, i in zip(self._model._param_list, self._model._param_plan_index) if i >= 0}
        L = _lib.lib()
        L.qt_adam_multi.argtypes = [ctypes.POINTER(AdamItem), ctypes.c_int, ctypes.POINTER(AdamDesc), ctypes.c_void_p]
        fused_groups = 0
        for group in self.param_groups:
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or p.device.type != "cuda" or not p.is_contiguous():
                    raise _lib.QtError("FusedAdam handles dense contiguous f32 parameters on the GPU")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                todo.append((p, st))
            if not todo:
                continue
            steps = {st["step"] for _, st in todo}
            for step in sorted(steps):   # parameters that joined later have their own bias correction
                part = [(p, st) for p, st in todo if st["step"] == step]
                desc = AdamDesc(group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                                1.0, step)
                in_plan = {plan_index[id(p)]: (p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"])
                           for p, st in part if id(p) in plan_index}
                rest = [(p, st) for p, st in part if id(p) not in plan_index]
                # one fused call per step() at most: the plan re-packs every operand copy inside it
                if in_plan and engine is not None and fused_groups == 0 and len(steps) == 1 and \
                        len(self.param_groups) == 1:
                    with torch.cuda.device(engine.device):
                        engine.adam_step(in_plan, desc)
                    fused_groups += 1
                else:
                    rest = part
                    if engine is not None:
                        engine.invalidate_weights()
                if rest:
                    items = (AdamItem * len(rest))()
                    keep = []
                    for j, (p, st) in enumerate(rest):
                        g = p.grad.contiguous()
                        keep.append(g)
                        items[j] = AdamItem(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                            st["exp_avg_sq"].data_ptr(), p.numel())
                    with torch.cuda.device(rest[0][0].device):
                        _lib.check(L.qt_adam_multi(items, len(rest), ctypes.byref(desc), _lib.stream_ptr()),
                                   "qt_adam_multi")
                    global _raw_updates
                    _raw_updates += 1
        return loss
