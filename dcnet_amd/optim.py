"""RMSprop with the step as one fused HIP pass per 32 tensors (dcn_rmsprop_step).

Same update, hyper-parameters and ``state_dict`` layout as ``torch.optim.RMSprop`` with ``momentum=0`` and
``centered=False`` — the optimiser the reference builds at train_DCNet.py:528-534 — so checkpoints move between the
two.  torch's foreach implementation makes five element-wise passes over parameters, gradients and state in ~20
launches (2.4 ms per step for the 74 M trained parameters of DCNet); this one reads and writes each value once.
"""
from __future__ import annotations

import ctypes

import torch

from .lib import lib


class RMSprop(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-2, alpha: float = 0.99, eps: float = 1e-8, weight_decay: float = 0.0,
                 momentum: float = 0.0, centered: bool = False):
        if momentum != 0.0 or centered:
            raise NotImplementedError("dcnet_amd.optim.RMSprop implements momentum=0, centered=False (the reference's setting)")
        if lr < 0 or eps < 0 or alpha < 0 or weight_decay < 0:
            raise ValueError("invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=0.0, centered=False))
        # device_lr: the step reads each group's learning rate from a device scalar (refreshed by sync_lr()) instead of a kernel
        # argument — what a step captured into a hipGraph needs to follow a schedule (dcnet_amd.graph.GraphedTrainStep)
        self.device_lr = False
        self._lr_dev = {}
        self._stepped = []           # the "step" counters touched by the last step() (bump_steps: replays of a captured step)

    LR_RING = 4

    def sync_lr(self, device=None) -> None:
        """Upload every group's current ``lr`` into its device scalar when it changed (async, from page-locked memory).

        The host may run several replays ahead of the GPU, so one staging word would be overwritten before its copy has
        run: each group stages through a ring of ``LR_RING`` pinned words, a slot is rewritten only after the event recorded
        behind its last copy has completed.  The "changed" test compares Python floats (the value last uploaded), not the
        float32 staging word against a double."""
        for gi, group in enumerate(self.param_groups):
            ent = self._lr_dev.get(gi)
            if ent is None:
                dev = device if device is not None else next(p.device for p in group["params"] if p.is_cuda)
                ent = {"pinned": torch.empty(self.LR_RING, dtype=torch.float32).pin_memory(), "dev": torch.empty(1, dtype=torch.float32, device=dev),
                       "events": [None] * self.LR_RING, "next": 0, "last": None}
                self._lr_dev[gi] = ent
            lr = float(group["lr"])
            if ent["last"] is not None and ent["last"] == lr:
                continue
            i = ent["next"]
            if ent["events"][i] is not None:
                ent["events"][i].synchronize()          # the copy that last read this slot has run
            ent["pinned"][i] = lr
            ent["dev"].copy_(ent["pinned"][i:i + 1], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(ent["dev"].device))
            ent["events"][i] = ev
            ent["next"] = (i + 1) % self.LR_RING
            ent["last"] = lr

    def bump_steps(self) -> None:
        """Advance the per-parameter ``step`` counters once more (a replay of a captured step ran the device update)."""
        if self._stepped:
            torch._foreach_add_(self._stepped, 1)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = lib()
        self._stepped = []
        if self.device_lr:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_lr()           # an eager step always sees the groups' current learning rates (a captured one: sync_lr() before the replay)
            elif not self._lr_dev:
                raise RuntimeError("dcnet_amd.optim.RMSprop: device_lr is set but sync_lr() was never called before the capture")
        for gi, group in enumerate(self.param_groups):
            ps, gs, vs, ns, keep = [], [], [], [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("dcnet_amd.optim.RMSprop: contiguous fp32 CUDA parameters only (no CPU path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                self._stepped.append(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ps.append(p.data_ptr()); gs.append(g.data_ptr()); vs.append(st["square_avg"].data_ptr()); ns.append(p.numel())
                keep.append(g)          # (a contiguous copy may be freed right after the launch: same-stream reuse is ordered)
            if not ps:
                continue
            n = len(ps)
            A = ctypes.c_void_p * n
            lr_dev = self._lr_dev[gi]["dev"].data_ptr() if self.device_lr else 0
            L.rmsprop_step(A(*ps), A(*gs), A(*vs), (ctypes.c_int64 * n)(*ns), n, float(group["lr"]), lr_dev, float(group["alpha"]),
                           float(group["eps"]), float(group["weight_decay"]), torch.cuda.current_stream().cuda_stream)
        return loss
