"""The opt-in device-side negative sampler (grounding_model.sampler = "device"; csrc/sample.hip dcn_device_sample; SURVEY H3 option (ii))
against the rules of the reference's random.sample loops (model/DCNet_model.py:62-96, 394-420): k DISTINCT positions, uniform over the
population, the matched / own position excluded — and against the exact (MT19937) sampler in expectation.  The numbers themselves are
not the reference's (documented: the default sampler is the bit-exact one)."""
import random

import numpy as np
import pytest
import torch

from util import build_product, synth_sd

pytestmark = pytest.mark.gpu


def _draw(dev, n, g0, seed, steps=1, top_k=30, neg_n=10, neg_c=5):
    from dcnet_amd.lib import lib
    hw = g0 * g0
    k9 = torch.zeros((n // 2, top_k, neg_n), dtype=torch.int64, device=dev)
    k14 = torch.zeros((n, hw, neg_c), dtype=torch.int64, device=dev)
    off = torch.zeros(hw + 1, dtype=torch.int32, device=dev)
    src = torch.zeros(n * hw * neg_c, dtype=torch.int32, device=dev)
    ws = torch.zeros(int(lib().device_sample_ws(hw)), dtype=torch.int32, device=dev)
    state = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
    out = []
    for _ in range(steps):
        lib().device_sample(state.data_ptr(), n, top_k, hw, neg_n, neg_c, k9.data_ptr(), k14.data_ptr(), off.data_ptr(), src.data_ptr(),
                            ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        out.append((k9.cpu().numpy().copy(), k14.cpu().numpy().copy(), off.cpu().numpy().copy(), src.cpu().numpy().copy()))
    return out, int(state.cpu()[1])


@pytest.mark.parametrize("n,g0", [(4, 8), (8, 13), (64, 13), (6, 19)])
def test_device_sampler_rules_and_inverse_table(dev, n, g0):
    """Ranges, distinctness, the excluded position, the inverse (CSR) table against the host's counting sort of the same draws,
    the step counter, and determinism in (seed, step)."""
    from dcnet_amd.lib import lib
    hw = g0 * g0
    draws, step = _draw(dev, n, g0, seed=1234, steps=3)
    assert step == 3
    for k9, k14, off, src in draws:
        assert k9.min() >= 0 and k9.max() < hw - 1                    # raw list positions of range(hw) minus the matched one
        assert (np.sort(k9, -1)[..., 1:] != np.sort(k9, -1)[..., :-1]).all()
        assert k14.min() >= 0 and k14.max() < hw
        assert (np.sort(k14, -1)[..., 1:] != np.sort(k14, -1)[..., :-1]).all()
        own = np.arange(hw)[:, None]
        assert (k14[n - 1] != own).all()                               # model/DCNet_model.py:84-88: its own position is removed for image N-1
        assert (k14[: n - 1] == own).any()                             # ... and only there
        ref_off = np.zeros(hw + 1, np.int32); ref_src = np.zeros(n * hw * 5, np.int32)
        k14c = np.ascontiguousarray(k14)
        lib().mt_sample_crossmodal_csr(k14c.ctypes.data, n, hw, 5, ref_off.ctypes.data, ref_src.ctypes.data)
        assert (off == ref_off).all() and (src == ref_src).all()
    assert not (draws[0][1] == draws[1][1]).all() and not (draws[0][0] == draws[1][0]).all()
    again, _ = _draw(dev, n, g0, seed=1234, steps=3)
    for a, b in zip(draws, again):
        assert all((x == y).all() for x, y in zip(a, b))
    other, _ = _draw(dev, n, g0, seed=1235, steps=1)
    assert not (other[0][1] == draws[0][1]).all()


def test_device_sampler_is_uniform(dev):
    """Chi-square of the drawn positions over 40 steps at the workload's grid (13 x 13, 64 images): K14 positions of the images in
    front (169 cells), of image N-1 (168 cells each row: pooled by offset from the own cell), K9 raw positions (168 cells); and the
    pair statistics a without-replacement draw must have (no value twice)."""
    from scipy.stats import chi2
    n, g0 = 64, 13
    hw = g0 * g0
    draws, _ = _draw(dev, n, g0, seed=77, steps=40)
    c14 = np.zeros(hw); c14_last = np.zeros(hw - 1); c9 = np.zeros(hw - 1)
    for k9, k14, _, _ in draws:
        c14 += np.bincount(k14[: n - 1].ravel(), minlength=hw)
        rel = (k14[n - 1] - np.arange(hw)[:, None]) % hw               # 1 .. hw-1 (never 0)
        c14_last += np.bincount(rel.ravel() - 1, minlength=hw - 1)
        c9 += np.bincount(k9.ravel(), minlength=hw - 1)
    for name, c in (("k14", c14), ("k14 last image", c14_last), ("k9", c9)):
        e = c.sum() / len(c)
        stat = float(((c - e) ** 2 / e).sum())
        lo, hi = chi2.ppf(1e-5, len(c) - 1), chi2.ppf(1 - 1e-5, len(c) - 1)
        assert lo < stat < hi, (name, stat, lo, hi)                    # neither skewed nor suspiciously flat


def test_device_sampler_losses_match_the_exact_sampler_in_expectation(dev):
    """The two contrastive losses (the only consumers of the draws) over 24 forwards with each sampler, same weights and inputs:
    the means agree within 2e-3 of the loss + four standard errors; every other output of the forward is bitwise the same."""
    from dcnet_amd import losses
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs
    size, n = 256, 4
    m = build_product(size, synth_sd(size), dev).train()
    image, word_id, word_mask = (t.to(dev) for t in synth_inputs(n, size, seed=5))
    bbox = synth_boxes(n, size, seed=5).to(dev)
    random.seed(3)
    stats = {}
    outs = {}
    for mode in ("mt", "device"):
        m.sampler = mode
        rec = []
        for _ in range(24):
            with torch.no_grad():
                out = m(image, word_id, word_mask)
                _, parts = losses.total_loss(out, bbox, size)
            rec.append([float(v) for v in parts.values()])
        stats[mode] = (list(parts.keys()), np.array(rec))
        outs[mode] = out
    keys, a = stats["mt"]; _, b = stats["device"]
    moved = 0
    for j, k_ in enumerate(keys):
        ma, mb = a[:, j].mean(), b[:, j].mean()
        se = np.sqrt(a[:, j].var(ddof=1) / len(a) + b[:, j].var(ddof=1) / len(b))
        assert abs(ma - mb) <= 2e-3 * max(1.0, abs(ma)) + 4 * se, (k_, ma, mb, se)
        moved += a[:, j].std() > 0
    assert moved >= 2, "no loss term depends on the draws?"
    st = random.getstate()
    m.sampler = "device"
    with torch.no_grad():
        m(image, word_id, word_mask)
    assert random.getstate() == st                                      # device mode leaves Python's stream alone
    for x, y in zip(outs["mt"][0], outs["device"][0]):
        assert torch.equal(x, y)                                        # outbox: untouched by the sampler


def test_device_sampler_in_the_captured_step(dev):
    """GraphedTrainStep with sampler = "device": no host draws, new negatives every replay, finite falling-in-range losses, and two
    runs from the same seed give the same losses bit for bit (the generator is keyed by (seed, step), nothing else)."""
    from dcnet_amd.graph import GraphedTrainStep
    from dcnet_amd.parallel import freeze_gradless
    from dcnet_amd.train import make_optimizer
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs
    size, n = 256, 4

    def run():
        m = build_product(size, synth_sd(size), dev)
        m.sampler = "device"; m.sampler_seed = 4242
        freeze_gradless(m)
        opt = make_optimizer(m, 1e-4)
        image, word_id, word_mask = (t.to(dev) for t in synth_inputs(n, size, seed=21))
        bbox = synth_boxes(n, size, seed=21).to(dev)
        st = random.getstate()
        step = GraphedTrainStep(m, opt, image, word_id, word_mask, bbox, size, warmup=1)
        negs, ls = [], []
        for _ in range(3):
            ls.append(float(step()))
            negs.append(m.last_choices["k14_neg"].clone())
        assert random.getstate() == st and step.host_draws is False and m.sampler_busy_s == 0.0
        assert not torch.equal(negs[0], negs[1]) and not torch.equal(negs[1], negs[2])
        return ls

    a, b = run(), run()
    assert a == b and all(np.isfinite(a))
