"""A training step of the hot path as ONE replayed hipGraph.

The reference's step is a Python loop body (train_DCNet.py:580-646: forward, five losses, ``backward()``, RMSprop).  Eagerly
that is ~1 300 kernel launches per step from one Python thread, which keeps that thread busy for most of the step's GPU time.
``GraphedTrainStep`` captures the same launches once — forward of ``grounding_model``, ``losses.total_loss``, autograd's
backward (with the weight-gradient / language / sampling side streams as forked branches of the graph) and the fused RMSprop
update — and replays them with one ``hipGraphLaunch`` per step.  What stays on the host per step:

  * the draws of the two sampling heads (Python's MT19937 stream must advance exactly as the reference's forward advances
    it): made natively on a worker thread, uploaded into static device buffers the captured kernels read
    (``grounding_model.draw_samples`` / ``static_samples``) — or, with ``grounding_model.sampler = "device"``, nothing: the draws
    are three captured kernels of a counter-based generator whose step counter lives on the device;
  * the learning rate of the schedule (train_DCNet.py:244-253): one device scalar per parameter group that the captured
    RMSprop kernel reads (``optim.RMSprop.device_lr``);
  * new input data: ``copy_`` into the static ``image`` / ``word_id`` / ``bbox`` tensors.

Data-parallel runs capture forward + backward only; the gradient all-reduce (one flat RCCL all-reduce,
``parallel.FlatGradAllReduce``) and the optimiser step follow the replay eagerly — no collective is captured.

Everything else is unchanged: the same kernels in the same order on the same streams, so a replayed step is bitwise equal to
an eager one (tests/test_graph_gpu.py).
"""
from __future__ import annotations

import time
from typing import Optional

import torch

from . import losses, ops


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, image, word_id, word_mask, bbox, size)`` then ``loss = step()`` per iteration.

    ``image``/``word_id``/``word_mask``/``bbox`` become the static inputs (``step.image.copy_(new)`` to feed new data of the same
    shape).  ``reducer``: a callable run after the replayed backward and before the optimiser step (data-parallel gradient
    averaging); with a reducer the optimiser step runs eagerly after it.  ``warmup`` eager steps are run first (they populate the
    library's caches: geometry tables, scratch buffers, function attributes) — they ARE training steps (parameters move)."""

    def __init__(self, model, optimizer, image, word_id, word_mask, bbox, size: int, reducer=None, warmup: int = 2):
        if not image.is_cuda:
            raise RuntimeError("GraphedTrainStep: HIP only (no CPU path)")
        self.model, self.opt, self.size, self.reducer = model, optimizer, size, reducer
        self.core = model.module if hasattr(model, "module") else model
        self._hooked = (self.core is not model or getattr(reducer, "uses_parameter_hooks", False)
                        or any(getattr(p_, "_post_accumulate_grad_hooks", None) for p_ in self.core.parameters()))
        dev = image.device
        self.image, self.word_id, self.bbox = image.clone(), word_id.clone(), bbox.clone()
        self.word_mask = None if word_mask is None else word_mask.clone()
        self.n = image.shape[0]
        self.samples = self.core.sample_buffers(self.n, dev)
        # grounding_model.sampler == "device": the draws are kernels of the captured forward (csrc/sample.hip dcn_device_sample, step
        # counter on the device) — nothing of the sampling heads is left on the host, Python's `random` stream is not advanced
        self.host_draws = getattr(self.core, "sampler", "mt") != "device"
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.loss = None
        self.parts = None
        self.replays = 0
        self.host_launch_s = 0.0         # host time spent launching (graph replay, reducer, optimiser step, learning-rate upload)
        self.host_sampler_s = 0.0        # host time spent waiting for / uploading the draws of the sampling heads (worker thread)
        self._done = [None, None]        # device sampler: events behind the last two replays (the host's throttle, __call__)
        if reducer is not None and hasattr(reducer, "check_bound_set"):
            reducer.check_bound_set(self.core)       # a bound flat buffer must not hold parameters autograd never writes
        if hasattr(optimizer, "device_lr"):
            optimizer.device_lr = True
            optimizer.sync_lr(dev)
        model.train()
        # warm-up on a side stream (torch.cuda.graph's documented recipe), then capture
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                if self.host_draws:
                    self.core.draw_samples(self.n, self.samples)
                self._body(eager=True)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self._zero_grads()
        if self.host_draws:
            self.core.draw_samples(self.n, self.samples)      # the draws of the captured pass (it runs once, as a real step)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread's calls are checked against the capture.  In a data-parallel run ProcessGroupNCCL's watchdog
        # thread polls the events of earlier collectives; under the default (global) mode one such hipEventQuery during the capture
        # aborts the process ("operation not permitted when stream is capturing": seen on a one-rank RCCL group, intermittently)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.loss, self.parts = self._body(eager=False)
        self.graph = g
        ops._amax_pools.pop(dev.index, None)                  # the capture's private abs-max pool stays with the graph
        self.replays = 0
        # the capture pass itself does not execute kernels: run it once so that state (parameters, running statistics) is that
        # of a step, with the draws made above
        self._replay_device()

    # ------------------------------------------------------------------------------------------------
    def _zero_grads(self):
        r = self.reducer
        if r is not None and getattr(r, "bound", False):
            r.zero()                         # gradients are views of the reducer's flat buffer: one memset, views stay attached
        else:
            self.opt.zero_grad(set_to_none=True)

    def _body(self, eager: bool):
        # the forward reads this step's draws from the static buffers (draw_samples filled them); only for this call — an eager
        # forward of the same model elsewhere keeps drawing for itself
        self.core.static_samples = self.samples
        defer = ops.LANGUAGE_BWD_DEFERRED and hasattr(self.core, "finish_backward")
        # direct = the head's conv blocks add their weight gradient to .grad themselves (autograd sees None): only where nobody listens
        # on the parameters — a DDP wrapper or a hook-based reducer would never see those gradients
        direct = ops.HEAD_WGRAD_DIRECT and not self._hooked
        was_direct = ops.WGRAD_DIRECT
        if defer:
            self.core.defer_language_backward = True
        ops.WGRAD_DIRECT = direct
        ops.reset_held_wgrads()
        try:
            out = self.model(self.image, self.word_id, self.word_mask)
            loss, parts = losses.total_loss(out, self.bbox, self.size)
            self._zero_grads()
            loss.backward()
            if defer:
                self.core.finish_backward()      # the language branch's backward, beside the backbone's (model.finish_backward)
            if direct:
                ops.finish_wgrads(self.image.device)     # the head's weight gradients went into .grad on the side stream (ops.WGRAD_DIRECT)
        finally:
            self.core.static_samples = None
            ops.WGRAD_DIRECT = was_direct
            if defer:
                self.core.defer_language_backward = False
        if self.reducer is None:
            self.opt.step()
        elif eager:
            self.reducer()
            self.opt.step()
        return loss.detach(), {k: v.detach() for k, v in parts.items()}

    def _replay_device(self):
        self.graph.replay()
        if self.reducer is not None:
            self.reducer()
            self.opt.step()
        elif hasattr(self.opt, "bump_steps") and self.replays > 0:
            self.opt.bump_steps()
        self.replays += 1

    def __call__(self):
        """One optimisation step.  Returns the (static) loss tensor of the step — reading it synchronises."""
        t0 = time.perf_counter()
        if hasattr(self.opt, "sync_lr"):
            self.opt.sync_lr()
        t1 = time.perf_counter()
        if self.host_draws:
            self.core.draw_samples(self.n, self.samples)      # (joins the worker thread; its staging set throttles the host to <= 2 steps ahead)
        else:
            # device sampler: nothing to join, so the host would run ahead until the runtime blocks it somewhere inside the launch.  The same
            # throttle, explicitly — wait for the replay of two steps ago — keeps the accounting honest: the wait is `host_sampler_s`
            # (bench.py: gpu_wait), the launch is the launch
            ev = self._done[self.replays & 1]
            if ev is not None:
                ev.synchronize()
        t2 = time.perf_counter()
        self._replay_device()
        if not self.host_draws:
            ev = torch.cuda.Event(); ev.record()
            self._done[(self.replays - 1) & 1] = ev
        t3 = time.perf_counter()
        self.host_launch_s += (t1 - t0) + (t3 - t2); self.host_sampler_s += t2 - t1
        return self.loss
