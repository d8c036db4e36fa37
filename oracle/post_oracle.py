"""CPU restatement of the reference's top-k cache and temporal post-processing.

TEST INFRASTRUCTURE ONLY — imported by tests/ and oracle/make_post_goldens.py, never by the product.
Pinned against the real reference by oracle/make_post_goldens.py (which drives the reference's own
``get_topk_pred_bbox`` and ``post_processing`` here and stores their outputs in
tests/golden/post_*.npz).

Follows, loop for loop:
  * ``topk_candidates``  test_DCNet.py:587-643 (save_cache body) + :662-705 (get_topk_pred_bbox)
  * ``temporal_fusion``  post_processing.py:230-284
  * ``letterbox_frame``  test_DCNet.py:615-625 / post_processing.py:300-311 (image extent after un-letterboxing)
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

ANCHORS_FULL = [(10., 13.), (16., 30.), (33., 23.), (30., 61.), (62., 45.), (59., 119.),
                (116., 90.), (156., 198.), (373., 326.)][::-1]                  # test_DCNet.py:142-147


def letterbox_frame(size: int, ratio: float, dw: float, dh: float) -> Tuple[int, int]:
    """(height, width) of the original image recovered from the letterboxed square:
    crop rows round(dh-0.1) .. size-round(dh+0.1), cols likewise, then resize by 1/ratio with Python
    ``round`` (test_DCNet.py:615-624)."""
    top, bottom = round(float(dh) - 0.1), size - round(float(dh) + 0.1)
    left, right = round(float(dw) - 0.1), size - round(float(dw) + 0.1)
    h, w = bottom - top, right - left
    return round(h / float(ratio)), round(w / float(ratio))


def topk_candidates(pred_anchor: Sequence[torch.Tensor], fvisu: Sequence[torch.Tensor], size: int, topk: int,
                    ratio: float, dw: float, dh: float, anchor_imsize: int = 416):
    """One clip (batch 1).  ``pred_anchor[s]``: (1,15,g,g) modulated head output, ``fvisu[s]``: (1,E,g,g).
    Returns (boxes (topk,1,4) in original-image pixels, scores list[topk], feats (topk,1,E),
    cells list of (scale, anchor, gj, gi))."""
    pa = [p.view(p.size(0), 3, 5, p.size(2), p.size(3)) for p in pred_anchor]  # :587-589
    conf_list = [p[:, :, 4, :, :].contiguous().view(1, -1) for p in pa]        # :595
    pred_conf = torch.cat(conf_list, dim=1)
    max_conf_topk, max_loc_topk = torch.topk(pred_conf, k=topk, dim=1)         # :602
    H, W = letterbox_frame(size, ratio, dw, dh)
    boxes, scores, feats, cells = [], [], [], []
    for ii in range(topk):                                                     # :627
        max_conf, max_loc = max_conf_topk[:, ii], max_loc_topk[:, ii]
        if max_loc[0] < 3 * (size // 32) ** 2:                                 # :668-673
            best_scale = 0
        elif max_loc[0] < 3 * (size // 32) ** 2 + 3 * (size // 16) ** 2:
            best_scale = 1
        else:
            best_scale = 2
        grid, grid_size = size // (32 // (2 ** best_scale)), 32 // (2 ** best_scale)
        anchors = [ANCHORS_FULL[x + 3 * best_scale] for x in (0, 1, 2)]
        scaled = [(x[0] / (anchor_imsize / grid), x[1] / (anchor_imsize / grid)) for x in anchors]
        pc = conf_list[best_scale].view(1, 3, grid, grid).numpy()
        (best_n, gj, gi) = np.where(pc[0] == max_conf.numpy()[0])              # :684 first exact match
        best_n, gi, gj = int(best_n[0]), int(gi[0]), int(gj[0])
        box = torch.zeros(1, 4)
        box[0, 0] = torch.sigmoid(pa[best_scale][0, best_n, 0, gj, gi]) + gi   # :690-694
        box[0, 1] = torch.sigmoid(pa[best_scale][0, best_n, 1, gj, gi]) + gj
        box[0, 2] = torch.exp(pa[best_scale][0, best_n, 2, gj, gi]) * scaled[best_n][0]
        box[0, 3] = torch.exp(pa[best_scale][0, best_n, 3, gj, gi]) * scaled[best_n][1]
        box = box * grid_size
        xy = torch.zeros(1, 4)                                                 # xywh2xyxy, utils/utils.py:34-40
        xy[:, 0] = box[:, 0] - box[:, 2] / 2; xy[:, 1] = box[:, 1] - box[:, 3] / 2
        xy[:, 2] = box[:, 0] + box[:, 2] / 2; xy[:, 3] = box[:, 1] + box[:, 3] / 2
        xy[:, 0], xy[:, 2] = (xy[:, 0] - dw) / ratio, (xy[:, 2] - dw) / ratio  # :698-699
        xy[:, 1], xy[:, 3] = (xy[:, 1] - dh) / ratio, (xy[:, 3] - dh) / ratio
        xy[:, :2] = torch.clamp(xy[:, :2], min=0)                              # :700-701
        xy[:, 2] = torch.clamp(xy[:, 2], max=W); xy[:, 3] = torch.clamp(xy[:, 3], max=H)
        boxes.append(xy); scores.append(float(max_conf[0]))
        feats.append(fvisu[best_scale][:, :, gj, gi])                          # :633
        cells.append((best_scale, best_n, gj, gi))
    return torch.stack(boxes), scores, torch.stack(feats), cells


def temporal_fusion(center_feat: torch.Tensor, ref_feats: List[torch.Tensor], ref_scores: List[torch.Tensor],
                    invalid: Sequence[int] = ()):
    """post_processing.py:246-278.  ``center_feat`` (topk,1,E); ``ref_feats[r]`` (topk,1,E) and
    ``ref_scores[r]`` (topk,) for each of the R frames of the window (centre included);
    ``invalid`` = window slots whose cache was missing.  Returns (index of the winning candidate,
    fused scores (topk,))."""
    topk = center_feat.shape[0]
    R = len(ref_feats)
    refer = torch.cat(ref_feats, dim=1)                                        # topk x R x E
    score = torch.stack(ref_scores).permute(1, 0)                              # topk x R
    E = refer.shape[2]
    refer = refer.view(-1, E).permute(1, 0)                                    # E x (topk*R)
    center = center_feat.unsqueeze(1).view(-1, E)
    sim = torch.bmm(center.unsqueeze(0), refer.unsqueeze(0))                   # 1 x topk x (topk*R)
    sim = sim.reshape(topk, topk, R)
    sim_max, sim_idx = sim.max(dim=1)                                          # best match per reference frame
    refer_score = score.gather(0, sim_idx)
    w = F.softmax(sim_max, dim=1)
    if len(invalid) > 0:
        w[:, list(invalid)] = 0
    fused = torch.sum(w * refer_score, dim=1)
    (idx,) = np.where(fused.numpy() == fused.max().numpy())
    return int(idx[0]), fused
