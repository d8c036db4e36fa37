"""Host logic that needs no GPU: the vectorised losses / target assignment / decode of dcnet_amd.losses
against the oracle's line-by-line restatement of train_DCNet.py, on CPU tensors."""
import random

import torch

from dcnet_amd import losses
from oracle import dcnet_oracle as O
from oracle import train_oracle as TO


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


def test_total_loss_matches_oracle():
    from dcnet_amd.utils.synth import synth_boxes
    for size, n, seed in ((256, 4, 0), (416, 6, 1), (256, 2, 2)):
        out = _fake_outputs(n, size, seed)
        bbox = synth_boxes(n, size, seed=seed)
        names = ["outbox", "sim_score", "loc_score", "corr_feat", "flang_attn", "frame_feature", "corrspendence_feature",
                 "neg_feature", "vit_posit", "lag_posit", "neg_cross"]
        l1, p1 = losses.total_loss(tuple(out[k] for k in names), bbox, size)
        l2, p2 = TO.total_loss(out, bbox, size)
        for k in p1:
            assert abs(float(p1[k]) - float(p2[k])) < 1e-5 * max(1.0, abs(float(p2[k]))), (k, float(p1[k]), float(p2[k]))
        assert abs(float(l1) - float(l2)) < 1e-4 * max(1.0, abs(float(l2)))


def test_build_target_matches_oracle():
    from dcnet_amd.utils.synth import synth_boxes
    for size in (256, 416):
        bbox = torch.clamp(synth_boxes(16, size, seed=size), 0, size - 1)
        b1, gi1, gj1, n1, c1 = losses.build_target(bbox, size)
        b2, gi2, gj2, n2, c2 = TO.build_target(bbox, size)
        assert n1.tolist() == n2 and gi1.tolist() == [int(x) for x in gi2] and gj1.tolist() == [int(x) for x in gj2]
        for a, b in zip(b1 + c1, b2 + c2):
            assert torch.allclose(a, b, atol=1e-6)


def test_decode_matches_oracle():
    for size in (256, 416):
        out = _fake_outputs(5, size, 3)["outbox"]
        a = losses.decode_boxes(out, size); b = O.decode_boxes(out, size)
        assert torch.allclose(a, b, atol=1e-3), (a - b).abs().max()
        assert torch.allclose(losses.bbox_iou(a, b), O.bbox_iou_xyxy(a, b))
