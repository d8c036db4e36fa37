"""The caller-side pieces of the hot path on the device (dcnet_amd.losses: target assignment, the five losses, box
decode, IoU) against the oracle's line-by-line restatement of train_DCNet.py:45-220,265-332,764-816 — the same three
comparisons as tests/test_losses_cpu.py, on device tensors (this is the path a training step runs), plus the gradients
of the total loss with respect to every model output."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fake_outputs(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    grids = [size // 32, size // 16, size // 8]
    r = lambda *s: torch.randn(*s, generator=g)
    out = dict(outbox=[r(n, 15, x, x) for x in grids], sim_score=[r(n, x, x) for x in grids],
               loc_score=[torch.rand(n, x, x, generator=g) for x in grids], corr_feat=[r(n, 512, x, x) for x in grids],
               flang_attn=r(n, 512, 1, 1),
               frame_feature=[r(n // 2, 512) for _ in range(30)], corrspendence_feature=[r(n // 2, 512) for _ in range(30)],
               neg_feature=[r(n // 2, 10, 512) for _ in range(30)],
               vit_posit=[r(n, 512) for _ in range(grids[0] ** 2)], lag_posit=[r(n, 1, 512) for _ in range(grids[0] ** 2)],
               neg_cross=[r(n, 5, 512) for _ in range(grids[0] ** 2)])
    return out

NAMES = ["outbox", "sim_score", "loc_score", "corr_feat", "flang_attn", "frame_feature", "corrspendence_feature",
         "neg_feature", "vit_posit", "lag_posit", "neg_cross"]


def _to(dev, v, grad=False):
    if isinstance(v, list):
        return [_to(dev, t, grad) for t in v]
    t = v.detach().to(dev)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("size,n,seed", [(256, 4, 0), (416, 6, 1), (256, 2, 2), (416, 16, 3)])
def test_total_loss_and_gradients_match_oracle_on_device(dev, size, n, seed):
    from dcnet_amd import losses
    from dcnet_amd.utils.synth import synth_boxes
    from oracle import train_oracle as TO
    out = _fake_outputs(n, size, seed)
    bbox = synth_boxes(n, size, seed=seed)
    ref_in = {k: _to("cpu", out[k], True) for k in NAMES}
    l2, p2 = TO.total_loss(ref_in, bbox, size)
    l2.backward()
    dev_in = {k: _to(dev, out[k], True) for k in NAMES}
    l1, p1 = losses.total_loss(tuple(dev_in[k] for k in NAMES), bbox.to(dev), size)
    l1.backward()
    for k in p1:
        assert abs(float(p1[k]) - float(p2[k])) < 2e-5 * max(1.0, abs(float(p2[k]))), (k, float(p1[k]), float(p2[k]))
    assert abs(float(l1) - float(l2)) < 1e-4 * max(1.0, abs(float(l2)))
    for k in NAMES:
        a, b = dev_in[k], ref_in[k]
        for x, y in zip(a if isinstance(a, list) else [a], b if isinstance(b, list) else [b]):
            if y.grad is None:
                assert x.grad is None or float(x.grad.abs().max()) == 0.0, k
                continue
            scale = max(float(y.grad.abs().max()), 1e-6)
            assert x.grad is not None, k
            assert float((x.grad.cpu() - y.grad).abs().max()) < 2e-4 * scale + 1e-7, (k, scale)


def test_build_target_matches_oracle_on_device(dev):
    from dcnet_amd import losses
    from dcnet_amd.utils.synth import synth_boxes
    from oracle import train_oracle as TO
    for size in (256, 416, 608):
        bbox = torch.clamp(synth_boxes(64, size, seed=size), 0, size - 1)
        b1, gi1, gj1, n1, c1 = losses.build_target(bbox.to(dev), size)
        b2, gi2, gj2, n2, c2 = TO.build_target(bbox, size)
        assert n1.tolist() == n2 and gi1.tolist() == [int(x) for x in gi2] and gj1.tolist() == [int(x) for x in gj2]
        for k, (a, b) in enumerate(zip(b1 + c1, b2 + c2)):
            d = float((a.cpu() - b).abs().max())
            assert d < 2e-5, (size, k, d)            # device logf vs host logf: a few ulp of values up to ~4


def test_decode_and_iou_match_oracle_and_reference_fixture_on_device(dev, golden_dir):
    import os
    import numpy as np
    from dcnet_amd import losses
    from oracle import dcnet_oracle as O
    for size in (256, 416):
        out = _fake_outputs(5, size, 3)["outbox"]
        a = losses.decode_boxes([t.to(dev) for t in out], size).cpu()
        b = O.decode_boxes(out, size)
        assert torch.allclose(a, b, atol=1e-3), (a - b).abs().max()
        assert torch.allclose(losses.bbox_iou(a.to(dev), b.to(dev)).cpu(), O.bbox_iou_xyxy(a, b), atol=1e-6)
    # the REFERENCE's decode (train_DCNet.py validate_epoch, driven by oracle/make_format_goldens.py) on stored inputs
    g = np.load(os.path.join(golden_dir, "decode_ref.npz"))
    for tag in ("256", "416"):
        outbox = [torch.from_numpy(g[f"outbox{s}_{tag}"]) for s in range(3)]
        ref_boxes = torch.from_numpy(g[f"pred_bbox_{tag}"]); gt = torch.from_numpy(g[f"gt_bbox_{tag}"])
        size = int(tag)
        got = losses.decode_boxes([t.to(dev) for t in outbox], size)
        assert float((got.cpu() - ref_boxes).abs().max()) < 1e-3
        iou = losses.bbox_iou(got, gt.to(dev)).cpu()
        assert torch.allclose(iou, torch.from_numpy(g[f"iou_{tag}"]), atol=1e-5)
        assert abs(float((iou > 0.5).float().mean()) - float(g[f"accu_{tag}"])) < 1e-6
