"""The captured training step (dcnet_amd.graph.GraphedTrainStep: forward + five losses + backward + RMSprop as one hipGraph)
against the eager step it was captured from: same kernels, same order, same streams -> the same bits.  The oracle-level parity of
the step itself is tests/test_model_gpu.py::test_train_forward_backward_matches_oracle; this file pins the replay to it."""
import random

import pytest
import torch

from util import build_product, synth_sd

pytestmark = pytest.mark.gpu


def _setup(dev, size, n, seed):
    from dcnet_amd.parallel import freeze_gradless
    from dcnet_amd.train import make_optimizer
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs
    m = build_product(size, synth_sd(size), dev)
    freeze_gradless(m)
    opt = make_optimizer(m, 1e-4)
    image, word_id, word_mask = (t.to(dev) for t in synth_inputs(n, size, seed=seed))
    bbox = synth_boxes(n, size, seed=seed).to(dev)
    return m, opt, image, word_id, word_mask, bbox


def test_replayed_steps_equal_eager_steps_bitwise(dev):
    """Five optimisation steps, once eagerly (train.train_step) and once as one eager warm-up step + the captured pass + three
    replays under a learning-rate schedule: identical losses, parameters, running statistics, optimiser state and Python RNG
    position."""
    from dcnet_amd.graph import GraphedTrainStep
    from dcnet_amd.train import adjust_learning_rate, train_step
    size, n, steps = 256, 4, 5
    lr_of = lambda it: 1e-4 if it < 2 else 1e-4 * (1 - it / 10.0)     # the constructor's two steps run at the initial rate
    m1, o1, image, word_id, word_mask, bbox = _setup(dev, size, n, 21)
    random.seed(99)
    ref_losses = []
    for it in range(steps):
        adjust_learning_rate(o1, 0, lr_of(it), 1, 0.9)
        loss, _ = train_step(m1, o1, image, word_id, word_mask, bbox, size)
        ref_losses.append(float(loss))
    rng_after = random.getstate()

    m2, o2, image, word_id, word_mask, bbox = _setup(dev, size, n, 21)
    random.seed(99)
    step = GraphedTrainStep(m2, o2, image, word_id, word_mask, bbox, size, warmup=1)      # steps 0 (eager) and 1 (captured pass)
    got = [None, float(step.loss)]
    for it in range(2, steps):
        adjust_learning_rate(o2, 0, lr_of(it), 1, 0.9)
        got.append(float(step()))
    assert got[1:] == ref_losses[1:], (got, ref_losses)
    assert random.getstate() == rng_after
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    s1, s2 = o1.state_dict()["state"], o2.state_dict()["state"]
    assert s1.keys() == s2.keys()
    for k in s1:
        assert torch.equal(s1[k]["square_avg"], s2[k]["square_avg"]), k
        assert float(s1[k]["step"]) == float(s2[k]["step"]) == steps, (k, float(s1[k]["step"]), float(s2[k]["step"]))


def test_new_inputs_reach_the_captured_step(dev):
    """copy_ into the static inputs changes what the next replay computes (loss equals an eager model's loss on that batch)."""
    from dcnet_amd import losses
    from dcnet_amd.graph import GraphedTrainStep
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs
    size, n = 256, 2
    m, opt, image, word_id, word_mask, bbox = _setup(dev, size, n, 5)
    for g in opt.param_groups:
        g["lr"] = 0.0                                   # parameters stay put: losses depend on the batch and the draws only
        g["weight_decay"] = 0.0
    random.seed(3)
    step = GraphedTrainStep(m, opt, image, word_id, word_mask, bbox, size, warmup=1)
    img2, wid2, _ = (t.to(dev) for t in synth_inputs(n, size, seed=6))
    box2 = synth_boxes(n, size, seed=6).to(dev)
    step.image.copy_(img2); step.word_id.copy_(wid2); step.bbox.copy_(box2)
    st = random.getstate()
    a = float(step())
    # the same batch, the same draws, eagerly on the same model (BatchNorm in train mode uses batch statistics; lr = 0)
    random.setstate(st)
    m.static_samples = None
    out = m(img2, wid2, word_mask)
    b = float(losses.total_loss(out, box2, size)[0])
    assert a == b, (a, b)


def test_graph_queues_of_the_captured_step(tmp_path):
    """The stream schedule of the captured step leans on how the hipGraph executor maps branches to its four queues (the k-th dependent of
    a node gets the node's queue + k; tools/graph_sched.py models it).  The runtime's own dump of the captured DAG
    (DEBUG_HIP_GRAPH_DOT_PRINT, in a child process: the flag is read when the runtime loads) must agree with that model on every node,
    and show what the schedule is built for: the backbone's data-gradient chain on ONE queue, its weight gradients on another, the
    language branch's backward on a third."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dot = str(tmp_path / "step.dot")
    env = dict(os.environ, DEBUG_HIP_GRAPH_DOT_PRINT="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_dot.py"), "--out", dot, "--clips", "2"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_sched.py"), dot, "--chains"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    c = json.loads(r.stdout.strip().splitlines()[-1])
    assert c["agree"] == c["nodes"] and c["nodes"] > 1000, c
    assert len(c["data_gradients"]) == 1, c
    main_q = next(iter(c["data_gradients"]))
    wq = max(c["weight_gradients"], key=c["weight_gradients"].get)
    assert wq != main_q and c["weight_gradients"][wq] >= 0.9 * sum(c["weight_gradients"].values()), c
    assert len(c["language"]) == 1 and next(iter(c["language"])) not in (main_q, wq), c


def test_staged_backward_gives_the_gradients_of_a_plain_backward(dev):
    """What GraphedTrainStep switches on for its step — the language branch's backward as its own stage (model.finish_backward) and the
    head blocks adding their weight gradients to .grad on the side stream (ops.WGRAD_DIRECT, ops.finish_wgrads) — run eagerly against
    a plain loss.backward() of the same model state, batch and draws: every gradient bitwise equal, none missing."""
    from dcnet_amd import losses, ops
    size, n = 256, 4
    m, _, image, word_id, word_mask, bbox = _setup(dev, size, n, 33)
    m.train()
    sd = {k: v.clone() for k, v in m.state_dict().items()}

    def grads(staged):
        m.load_state_dict(sd)
        for p in m.parameters():
            p.grad = None
        random.seed(7)
        was = ops.WGRAD_DIRECT
        m.defer_language_backward = staged
        ops.WGRAD_DIRECT = staged
        try:
            out = m(image, word_id, word_mask)
            loss, _ = losses.total_loss(out, bbox, size)
            loss.backward()
            if staged:
                m.finish_backward()
                ops.finish_wgrads(dev)
        finally:
            ops.WGRAD_DIRECT = was
            m.defer_language_backward = False
        torch.cuda.synchronize()
        return float(loss), {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()}

    l0, g0 = grads(False)
    l1, g1 = grads(True)
    assert l0 == l1
    assert sum(v is not None for v in g0.values()) > 100
    for k in g0:
        assert (g0[k] is None) == (g1[k] is None), k
        if g0[k] is not None:
            assert torch.equal(g0[k], g1[k]), k
