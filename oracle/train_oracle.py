"""CPU oracle for the caller-side training pieces that drive backward through the
hot path: target assignment and the five losses of train_DCNet.py.

TEST INFRASTRUCTURE ONLY (see oracle/dcnet_oracle.py header).  Parity status:
PINNED by oracle/make_goldens.py against the imported reference ``train_DCNet``
module (loss scalars + gradient norms, tests/golden/train_*.npz).

All functions are device-agnostic torch code, so the GPU parity tests can apply
the very same loss to the HIP model's outputs and to the oracle model's outputs.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .dcnet_oracle import ANCHORS_FULL, bbox_iou_xyxy

Tensor = torch.Tensor


def build_target(raw_coord: Tensor, size: int, anchor_imsize: int = 416):
    """train_DCNet.py:265-332.  raw_coord (N,4) xyxy in pixels (already clamped to
    [0,size-1], :605).  Returns (bbox_list[3] (N,3,5,g,g), gi, gj, best_n_list,
    bbox_center_list[3] (N,5,g,g)) on raw_coord's device."""
    dev = raw_coord.device
    N = raw_coord.shape[0]
    coord_list, bbox_list, center_list = [], [], []
    for s in range(3):
        grid = size // (32 // (2 ** s))
        c = torch.stack([(raw_coord[:, 0] + raw_coord[:, 2]) / (2 * size),
                         (raw_coord[:, 1] + raw_coord[:, 3]) / (2 * size),
                         (raw_coord[:, 2] - raw_coord[:, 0]) / size,
                         (raw_coord[:, 3] - raw_coord[:, 1]) / size], 1) * grid     # :270-274
        coord_list.append(c)
        bbox_list.append(torch.zeros(N, 3, 5, grid, grid, device=dev))
        center_list.append(torch.zeros(N, 5, grid, grid, device=dev))
    best_n_list, best_gi, best_gj = [], [], []
    for ii in range(N):
        ious = []
        for s in range(3):
            grid = size // (32 // (2 ** s))
            gw, gh = coord_list[s][ii, 2], coord_list[s][ii, 3]
            anchors = [(a[0] / (anchor_imsize / grid), a[1] / (anchor_imsize / grid))
                       for a in ANCHORS_FULL[3 * s:3 * s + 3]]                      # :293-296
            gt = torch.tensor([[0., 0., float(gw), float(gh)]])
            an = torch.tensor([[0., 0., a[0], a[1]] for a in anchors], dtype=torch.float32)
            ious += [float(v) for v in bbox_iou_xyxy(gt, an)]                       # :298-303
        best_n = int(np.argmax(np.array(ious)))                                     # :305
        bs = best_n // 3
        grid = size // (32 / (2 ** bs))                                             # float, :308
        anchors = [(a[0] / (anchor_imsize / grid), a[1] / (anchor_imsize / grid))
                   for a in ANCHORS_FULL[3 * bs:3 * bs + 3]]
        gi = coord_list[bs][ii, 0].long(); gj = coord_list[bs][ii, 1].long()
        tx = coord_list[bs][ii, 0] - gi.float(); ty = coord_list[bs][ii, 1] - gj.float()
        gw, gh = coord_list[bs][ii, 2], coord_list[bs][ii, 3]
        tw = torch.log(gw / anchors[best_n % 3][0] + 1e-16)
        th = torch.log(gh / anchors[best_n % 3][1] + 1e-16)
        v = torch.stack([tx, ty, tw, th, torch.ones((), device=dev)])
        bbox_list[bs][ii, best_n % 3, :, gj, gi] = v                                # :322
        center_list[bs][ii, :, gj, gi] = v                                          # :323
        best_n_list.append(best_n); best_gi.append(gi); best_gj.append(gj)
    return bbox_list, best_gi, best_gj, best_n_list, center_list


def yolo_loss(inp: Sequence[Tensor], target: Sequence[Tensor], gi, gj, best_n_list, w_coord: float = 5.):
    """train_DCNet.py:45-72.  inp[s] (N,3,5,g,g)."""
    N = inp[0].size(0)
    pb, gb = [], []
    for ii in range(N):
        s, a = best_n_list[ii] // 3, best_n_list[ii] % 3
        t = inp[s][ii, a, :, gj[ii], gi[ii]]
        pb.append(torch.cat([torch.sigmoid(t[0:2]), t[2:4]]))
        gb.append(target[s][ii, a, :4, gj[ii], gi[ii]])
    pb = torch.stack(pb); gb = torch.stack(gb)
    l = sum(F.mse_loss(pb[:, k], gb[:, k]) for k in range(4))
    pred_conf = torch.cat([x[:, :, 4].reshape(N, -1) for x in inp], dim=1)
    gt_conf = torch.cat([x[:, :, 4].reshape(N, -1) for x in target], dim=1)
    return l * w_coord + F.cross_entropy(pred_conf, gt_conf.max(1)[1])


def _contrastive(q: Tensor, pos: Tensor, neg: Tensor, T: float) -> Tensor:
    l_pos = torch.einsum("nc,nc->n", q, pos).unsqueeze(-1)
    l_neg = torch.einsum("nc,nck->nk", q, neg)
    logits = torch.cat([l_pos, l_neg], dim=1) / T
    return F.cross_entropy(logits, torch.zeros(logits.shape[0], dtype=torch.long, device=q.device))


def interframe_contrastive_loss(q_list, k_list, neg_list, T: float = 0.07) -> Tensor:
    """train_DCNet.py:114-136."""
    loss = 0
    for q, k, neg in zip(q_list, k_list, neg_list):
        q = F.normalize(q, dim=1); k = F.normalize(k, dim=1)
        neg = F.normalize(neg.permute(0, 2, 1), dim=1)
        loss = _contrastive(q, k, neg, T) + loss
    return loss / len(q_list)


def crossmodal_contrastive_loss(q_list, k_list, neg_list, T: float = 0.07) -> Tensor:
    """train_DCNet.py:140-166 (k is NOT normalised as a whole; each k[:,jj] is)."""
    loss = 0
    for q, k, neg in zip(q_list, k_list, neg_list):
        q = F.normalize(q, dim=1)
        neg = F.normalize(neg.permute(0, 2, 1), dim=1)
        tmp = 0
        for jj in range(k.shape[1]):
            tmp = _contrastive(q, F.normalize(k[:, jj, :], dim=1), neg, T) + tmp
        loss = loss + tmp * 1.0 / k.shape[1]
    return loss / len(q_list)


def rank_loss(sim_score, neg_sim_score, target_center, margin: float = 0.1) -> Tensor:
    """train_DCNet.py:173-203."""
    N = sim_score[0].size(0)
    pos = torch.cat([s.reshape(N, -1) for s in sim_score], dim=1)
    neg = torch.cat([s.reshape(N, -1) for s in neg_sim_score], dim=1)
    gt = torch.cat([t[:, 4].reshape(N, -1) for t in target_center], dim=1)
    pos_p = (pos * gt).sum(-1)
    neg1 = (neg * gt).sum(-1)
    neg2 = (pos * gt.flip(0)).sum(-1)                                   # :197-198
    loss = torch.clamp(margin + neg1 - pos_p, 0) + torch.clamp(margin + neg2 - pos_p, 0)
    return loss.sum() / (N * 2)


def loc_loss(loc_score, target_center) -> Tensor:
    """train_DCNet.py:205-220."""
    N = loc_score[0].size(0)
    loc = torch.cat([s.reshape(N, -1) for s in loc_score], dim=1)
    gt = torch.cat([t[:, 4].reshape(N, -1) for t in target_center], dim=1)
    return F.cross_entropy(loc, gt.max(1)[1])


def total_loss(out: dict, bbox: Tensor, size: int):
    """train_DCNet.py:613-642 applied to the 11 outputs of the train forward
    (``out`` uses the key names of dcnet_oracle.grounding_forward_pairs).
    Returns (loss, dict of the five scalars)."""
    bbox = torch.clamp(bbox, min=0, max=size - 1)                        # :605
    pred = out["outbox"]
    gt_param, gi, gj, best_n, gt_center = build_target(bbox, size)
    pred5 = [p.view(p.size(0), 3, 5, p.size(2), p.size(3)) for p in pred]           # :618-620
    fa = out["flang_attn"]
    neg_sim = [torch.sum(fa.flip(0) * cf[:, :512], dim=1) for cf in out["corr_feat"]]   # :623-627
    l_yolo = yolo_loss(pred5, gt_param, gi, gj, best_n)
    l_rank = rank_loss(out["sim_score"], neg_sim, gt_center)
    l_inter = interframe_contrastive_loss(out["frame_feature"], out["corrspendence_feature"], out["neg_feature"])
    l_cross = crossmodal_contrastive_loss(out["vit_posit"], out["lag_posit"], out["neg_cross"])
    l_loc = loc_loss(out["loc_score"], gt_center)
    loss = l_yolo + 100 * l_rank + l_loc + 100 * l_inter + l_cross                  # :642
    return loss, dict(yolo=l_yolo, rank=l_rank, interframe=l_inter, cross=l_cross, loc=l_loc)
