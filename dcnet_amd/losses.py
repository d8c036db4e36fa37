"""Target assignment and the five training losses of DCNet, vectorised on the device.

The reference computes these on the caller side (train_DCNet.py:45-220 losses, :265-332
build_target, :613-642 combination) with per-sample Python loops and ``.item()`` host syncs.
Here every step is a batched tensor op, so one training step issues no host synchronisation
between forward and backward.  Semantics follow the reference line by line; the CPU restatement
used by the tests lives in oracle/train_oracle.py.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn.functional as F

ANCHORS_FULL = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119),
                (116, 90), (156, 198), (373, 326)][::-1]          # train_DCNet.py:404-406 (reversed)

_const_cache = {}


# Set by grounding_model.forward (train mode): the stream its sampling heads ran on, or None.  total_loss evaluates the
# two contrastive losses there; every consumer on the current stream is ordered behind it by an explicit wait.
CONTRASTIVE_STREAM = None


def _const(key, device, builder):
    """Small constant tensors, uploaded once per (key, device): a ``torch.tensor(list, device=...)`` in the
    step would be a pageable H2D copy that blocks the host until the queued forward kernels have drained."""
    k = (key, str(device))
    t = _const_cache.get(k)
    if t is None:
        t = builder().to(device)
        _const_cache[k] = t
    return t


def build_target(raw_coord: torch.Tensor, size: int, anchor_imsize: int = 416):
    """train_DCNet.py:265-332, batched.  raw_coord (N,4) xyxy pixels.  Returns
    (bbox_list[3] (N,3,5,g,g), gi (N,), gj (N,), best_n (N,), bbox_center_list[3] (N,5,g,g))."""
    dev = raw_coord.device
    N = raw_coord.shape[0]
    grids = [size // (32 // (2 ** s)) for s in range(3)]
    base = torch.stack([(raw_coord[:, 0] + raw_coord[:, 2]) / (2 * size), (raw_coord[:, 1] + raw_coord[:, 3]) / (2 * size),
                        (raw_coord[:, 2] - raw_coord[:, 0]) / size, (raw_coord[:, 3] - raw_coord[:, 1]) / size], 1)
    coords = [base * g for g in grids]                                                     # :270-274
    ious = []
    for s, g in enumerate(grids):
        anc = _const(("anc", s, g, anchor_imsize), dev, lambda: torch.tensor(
            [(a[0] / (anchor_imsize / g), a[1] / (anchor_imsize / g)) for a in ANCHORS_FULL[3 * s:3 * s + 3]],
            dtype=torch.float32))                                                          # (3,2)
        gw, gh = coords[s][:, 2:3], coords[s][:, 3:4]
        inter = torch.clamp(torch.min(gw, anc[None, :, 0]), min=0) * torch.clamp(torch.min(gh, anc[None, :, 1]), min=0)
        ious.append(inter / (gw * gh + anc[None, :, 0] * anc[None, :, 1] - inter + 1e-16))   # utils.bbox_iou :76-104
    best_n = torch.cat(ious, dim=1).argmax(dim=1)                                          # first max, like np.argmax :305
    best_scale = best_n // 3
    ar = torch.arange(N, device=dev)
    cs = torch.stack(coords, 0)[best_scale, ar]                                            # (N,4) at the best scale
    gi, gj = cs[:, 0].long(), cs[:, 1].long()
    anc_all = _const("anc_all", dev, lambda: torch.tensor(ANCHORS_FULL, dtype=torch.float32))   # (9,2)
    gsel = _const(("gridsf", tuple(grids)), dev, lambda: torch.tensor(grids, dtype=torch.float32))[best_scale]
    sa = anc_all[best_n] / (anchor_imsize / gsel).unsqueeze(1)                             # scaled anchor (N,2)
    tvec = torch.stack([cs[:, 0] - gi.float(), cs[:, 1] - gj.float(),
                        torch.log(cs[:, 2] / sa[:, 0] + 1e-16), torch.log(cs[:, 3] / sa[:, 1] + 1e-16),
                        torch.ones(N, device=dev)], 1)                                     # :314-322
    bbox_list, center_list = [], []
    for s, g in enumerate(grids):
        # samples whose best scale is not s write a zero vector into cell (0,0) of anchor 0 of THEIR OWN row n, which is
        # all zeros at this scale: every index tuple starts with the sample's n, so no two writes collide and a plain
        # scatter (accumulate=False) is exact — the accumulate form sorts its indices and cost 2 ms per step
        m = (best_scale == s).float().unsqueeze(1)
        a_s = torch.where(best_scale == s, best_n % 3, torch.zeros_like(best_n))
        gj_s = torch.where(best_scale == s, gj, torch.zeros_like(gj)); gi_s = torch.where(best_scale == s, gi, torch.zeros_like(gi))
        b = torch.zeros(N, 3, 5, g, g, device=dev); c = torch.zeros(N, 5, g, g, device=dev)
        k5 = _const("k5", dev, lambda: torch.arange(5)).unsqueeze(0)
        b.index_put_((ar.unsqueeze(1), a_s.unsqueeze(1), k5, gj_s.unsqueeze(1), gi_s.unsqueeze(1)), tvec * m, accumulate=False)
        c.index_put_((ar.unsqueeze(1), k5, gj_s.unsqueeze(1), gi_s.unsqueeze(1)), tvec * m, accumulate=False)
        bbox_list.append(b); center_list.append(c)
    return bbox_list, gi, gj, best_n, center_list


def _flat_index(best_n, gi, gj, grids, with_anchor: bool):
    """Index of each sample's positive cell in the scale-concatenated, flattened map."""
    dev = best_n.device
    g = _const(("gridsl", tuple(grids)), dev, lambda: torch.tensor(grids))[best_n // 3]
    mult = 3 if with_anchor else 1
    off = _const(("off", tuple(grids), mult), dev, lambda: torch.tensor(
        [sum(mult * x * x for x in grids[:i]) for i in range(len(grids))]))
    a = (best_n % 3) if with_anchor else torch.zeros_like(best_n)
    return off[best_n // 3] + a * g * g + gj * g + gi


def yolo_loss(pred5: Sequence[torch.Tensor], target: Sequence[torch.Tensor], gi, gj, best_n, w_coord: float = 5.):
    """train_DCNet.py:45-72.  pred5[s] (N,3,5,g,g)."""
    N = pred5[0].size(0)
    dev = pred5[0].device
    ar = torch.arange(N, device=dev)
    pb = torch.zeros(N, 4, device=dev); gb = torch.zeros(N, 4, device=dev)
    for s in range(3):
        m = (best_n // 3) == s
        g = pred5[s].shape[-1]
        a_s = torch.where(m, best_n % 3, torch.zeros_like(best_n))
        gj_s = torch.where(m, gj, torch.zeros_like(gj)); gi_s = torch.where(m, gi, torch.zeros_like(gi))
        t = pred5[s][ar, a_s, :, gj_s, gi_s]                                              # (N,5); rows of other scales masked below
        mf = m.float().unsqueeze(1)
        pb = pb + mf * torch.cat([torch.sigmoid(t[:, 0:2]), t[:, 2:4]], 1)
        gb = gb + mf * target[s][ar, a_s, :4, gj_s, gi_s]
    l = sum(F.mse_loss(pb[:, k], gb[:, k]) for k in range(4))
    pred_conf = torch.cat([x[:, :, 4].reshape(N, -1) for x in pred5], dim=1)
    grids = [x.shape[-1] for x in pred5]
    return l * w_coord + F.cross_entropy(pred_conf, _flat_index(best_n, gi, gj, grids, True))


def _contrastive(q, pos, neg, T):
    """q (K,n,c), pos (K,n,c), neg (K,n,c,m): mean over K of CE([q.pos, q.neg]/T, 0)."""
    l_pos = (q * pos).sum(-1, keepdim=True)
    l_neg = torch.einsum("knc,kncm->knm", q, neg)
    logits = torch.cat([l_pos, l_neg], dim=2) / T
    return F.cross_entropy(logits.flatten(0, 1), torch.zeros(logits.shape[0] * logits.shape[1], dtype=torch.long, device=q.device))


def interframe_contrastive_loss(q_list, k_list, neg_list, T: float = 0.07):
    """train_DCNet.py:114-136 (lists of equal-shape tensors -> one batched evaluation)."""
    q = F.normalize(torch.stack(list(q_list)), dim=2)
    k = F.normalize(torch.stack(list(k_list)), dim=2)
    neg = F.normalize(torch.stack(list(neg_list)).permute(0, 1, 3, 2), dim=2)
    return _contrastive(q, k, neg, T)


def crossmodal_contrastive_loss(q_list, k_list, neg_list, T: float = 0.07):
    """train_DCNet.py:140-166."""
    q = F.normalize(torch.stack(list(q_list)), dim=2)
    k = torch.stack(list(k_list))                                                          # (K,n,J,c)
    neg = F.normalize(torch.stack(list(neg_list)).permute(0, 1, 3, 2), dim=2)
    loss = 0
    for jj in range(k.shape[2]):
        loss = loss + _contrastive(q, F.normalize(k[:, :, jj, :], dim=2), neg, T)
    return loss / k.shape[2]


def rank_loss(sim_score, neg_sim_score, target_center, margin: float = 0.1):
    """train_DCNet.py:173-203."""
    N = sim_score[0].size(0)
    pos = torch.cat([s.reshape(N, -1) for s in sim_score], dim=1)
    neg = torch.cat([s.reshape(N, -1) for s in neg_sim_score], dim=1)
    gt = torch.cat([t[:, 4].reshape(N, -1) for t in target_center], dim=1)
    pos_p = (pos * gt).sum(-1)
    loss = torch.clamp(margin + (neg * gt).sum(-1) - pos_p, 0) + torch.clamp(margin + (pos * gt.flip(0)).sum(-1) - pos_p, 0)
    return loss.sum() / (N * 2)


def loc_loss(loc_score, best_n, gi, gj):
    """train_DCNet.py:205-220."""
    N = loc_score[0].size(0)
    loc = torch.cat([s.reshape(N, -1) for s in loc_score], dim=1)
    return F.cross_entropy(loc, _flat_index(best_n, gi, gj, [s.shape[-1] for s in loc_score], False))


def total_loss(outputs, bbox: torch.Tensor, size: int):
    """train_DCNet.py:605-642 on the 11-tuple of grounding_model.forward (train mode).
    Returns (loss, dict of the five parts)."""
    (pred, sim, loc, corr_feat, flang_attn, frame_f, corr_f, neg_f, vit_p, lag_p, neg_c) = outputs
    bbox = torch.clamp(bbox, min=0, max=size - 1)
    gt_param, gi, gj, best_n, gt_center = build_target(bbox, size)
    pred5 = [p.view(p.size(0), 3, 5, p.size(2), p.size(3)) for p in pred]
    neg_sim = [torch.sum(flang_attn.flip(0) * cf[:, :512], dim=1) for cf in corr_feat]      # :623-627
    cs = CONTRASTIVE_STREAM if frame_f[0].is_cuda else None
    if cs is not None:
        # the sampled lists were produced on the model's sampling stream: evaluate their two losses there as well, so
        # that autograd replays this whole branch (gathers, top-k bookkeeping, normalisations) beside the heads' backward
        main = torch.cuda.current_stream()
        with torch.cuda.stream(cs):
            inter = interframe_contrastive_loss(frame_f, corr_f, neg_f)
            cross = crossmodal_contrastive_loss(vit_p, lag_p, neg_c)
        main.wait_stream(cs)
        inter.record_stream(main); cross.record_stream(main)
    else:
        inter = interframe_contrastive_loss(frame_f, corr_f, neg_f)
        cross = crossmodal_contrastive_loss(vit_p, lag_p, neg_c)
    parts = dict(yolo=yolo_loss(pred5, gt_param, gi, gj, best_n),
                 rank=rank_loss(sim, neg_sim, gt_center),
                 interframe=inter,
                 cross=cross,
                 loc=loc_loss(loc, best_n, gi, gj))
    loss = parts["yolo"] + 100 * parts["rank"] + parts["loc"] + 100 * parts["interframe"] + parts["cross"]   # :642
    return loss, parts


def decode_boxes(outbox: List[torch.Tensor], size: int, anchor_imsize: int = 416) -> torch.Tensor:
    """Evaluation decode (train_DCNet.py:764-810), batched: global arg-max of the modulated
    confidence over 3 scales x 3 anchors, then (sigmoid(tx)+gi, sigmoid(ty)+gj, exp(tw)*aw,
    exp(th)*ah)*stride and xywh -> xyxy."""
    N = outbox[0].shape[0]
    dev = outbox[0].device
    ob = [o.view(N, 3, 5, o.shape[2], o.shape[3]) for o in outbox]
    grids = [o.shape[-1] for o in outbox]
    conf = torch.cat([o[:, :, 4].reshape(N, -1) for o in ob], dim=1)
    loc = conf.argmax(dim=1)
    off = _const(("off", tuple(grids), 3), dev, lambda: torch.tensor([sum(3 * x * x for x in grids[:i]) for i in range(len(grids))]))
    sc = (loc.unsqueeze(1) >= off.unsqueeze(0)).sum(1) - 1
    g = _const(("gridsl", tuple(grids)), dev, lambda: torch.tensor(grids))[sc]
    l = loc - off[sc]
    a = l // (g * g); gj = (l % (g * g)) // g; gi = l % g
    ar = torch.arange(N, device=dev)
    t = torch.zeros(N, 4, device=dev)
    for s in range(3):
        m = sc == s
        z = torch.zeros_like(a)
        t = t + m.float().unsqueeze(1) * ob[s][ar, torch.where(m, a, z), :4, torch.where(m, gj, z), torch.where(m, gi, z)]
    anc = _const("anc_all", dev, lambda: torch.tensor(ANCHORS_FULL, dtype=torch.float32))[sc * 3 + a] / (anchor_imsize / g.float()).unsqueeze(1)
    stride = (size // g).float()
    x = (torch.sigmoid(t[:, 0]) + gi) * stride; y = (torch.sigmoid(t[:, 1]) + gj) * stride
    w = torch.exp(t[:, 2]) * anc[:, 0] * stride; h = torch.exp(t[:, 3]) * anc[:, 1] * stride
    return torch.stack([x - w / 2, y - h / 2, x + w / 2, y + h / 2], 1)


def bbox_iou(box1: torch.Tensor, box2: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:76-104 (x1y1x2y2)."""
    iw = torch.clamp(torch.min(box1[:, 2], box2[:, 2]) - torch.max(box1[:, 0], box2[:, 0]), 0)
    ih = torch.clamp(torch.min(box1[:, 3], box2[:, 3]) - torch.max(box1[:, 1], box2[:, 1]), 0)
    inter = iw * ih
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1]); a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    return inter / (a1 + a2 - inter + 1e-16)
