"""Top-k candidate cache and temporal post-processing on the device.

The reference does this one clip at a time with host loops and ``.cpu().numpy()`` round trips
(test_DCNet.py:546-705 ``save_cache`` / ``get_topk_pred_bbox``; post_processing.py:181-284).  Here both
stages are batched tensor programs without host synchronisation:

  * ``topk_candidates``   top-k of the modulated confidence over 3 scales x 3 anchors, box decode,
                          un-letterbox + clamp, and the 512-d correspondence feature of each winning cell
  * ``temporal_fusion``   candidate-to-candidate similarity against every frame of the window, max over the
                          reference frame's candidates, softmax over frames, score fusion, arg-max
  * ``save_cache_entry`` / ``load_cache_entry`` / ``fuse_from_cache``   the reference's on-disk cache
                          format (``pred_bbox_topk`` (k,1,4), ``pred_score_topk`` list, ``visu_feat`` (k,1,E))
                          and its missing-neighbour rule (fall back to the centre entry, weight zeroed)
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .losses import ANCHORS_FULL, _const


def letterbox_frame(size: int, ratio: float, dw: float, dh: float) -> Tuple[int, int]:
    """(height, width) of the un-letterboxed image, with the reference's rounding (test_DCNet.py:615-624)."""
    top, bottom = round(float(dh) - 0.1), size - round(float(dh) + 0.1)
    left, right = round(float(dw) - 0.1), size - round(float(dw) + 0.1)
    return round((bottom - top) / float(ratio)), round((right - left) / float(ratio))


def topk_candidates(outbox: Sequence[torch.Tensor], corr_feat: Sequence[torch.Tensor], size: int, topk: int,
                    ratio: torch.Tensor, dw: torch.Tensor, dh: torch.Tensor, frame_hw: torch.Tensor,
                    anchor_imsize: int = 416):
    """Batched ``save_cache`` core (test_DCNet.py:587-643, :662-705).

    outbox[s]    (B,15,g,g) modulated head output of the n_frame model
    corr_feat[s] (B,E,g,g)  its correspondence features (any strides)
    ratio, dw, dh (B,)      letterbox meta;  frame_hw (B,2) = un-letterboxed (height, width)
    Returns boxes (B,k,4) xyxy in original-image pixels, scores (B,k), feats (B,k,E),
    cells (B,k,4) int64 = (scale, anchor, gj, gi).  Ties in the confidence resolve to torch.topk's
    choice (the reference takes the first exact match of the value, :684)."""
    B = outbox[0].shape[0]
    dev = outbox[0].device
    ob = [o.reshape(B, 3, 5, o.shape[2], o.shape[3]) for o in outbox]
    grids = [o.shape[-1] for o in outbox]
    conf = torch.cat([o[:, :, 4].reshape(B, -1) for o in ob], dim=1)
    score, loc = torch.topk(conf, k=topk, dim=1)                               # (B,k)
    off = _const(("off", tuple(grids), 3), dev,
                 lambda: torch.tensor([sum(3 * x * x for x in grids[:i]) for i in range(len(grids))]))
    sc = (loc.unsqueeze(2) >= off.view(1, 1, -1)).sum(2) - 1                   # scale of each winner
    g = _const(("gridsl", tuple(grids)), dev, lambda: torch.tensor(grids))[sc]
    l = loc - off[sc]
    a = l // (g * g); gj = (l % (g * g)) // g; gi = l % g
    bi = torch.arange(B, device=dev).unsqueeze(1).expand(B, topk)
    E = corr_feat[0].shape[1]
    t = torch.zeros(B, topk, 4, device=dev)
    feats = torch.zeros(B, topk, E, device=dev)
    zero = torch.zeros_like(a)
    for s in range(3):
        m = sc == s
        aa, jj, ii = torch.where(m, a, zero), torch.where(m, gj, zero), torch.where(m, gi, zero)
        mf = m.unsqueeze(2).float()
        t = t + mf * ob[s][bi, aa, :4, jj, ii]
        feats = feats + mf * corr_feat[s][bi, :, jj, ii]
    gf = g.float()
    anc = _const("anc_all", dev, lambda: torch.tensor(ANCHORS_FULL, dtype=torch.float32))[sc * 3 + a] \
        / (anchor_imsize / gf).unsqueeze(2)
    stride = (32 // (2 ** sc)).float()                                         # grid_size, :675
    x = (torch.sigmoid(t[..., 0]) + gi) * stride; y = (torch.sigmoid(t[..., 1]) + gj) * stride
    w = torch.exp(t[..., 2]) * anc[..., 0] * stride; h = torch.exp(t[..., 3]) * anc[..., 1] * stride
    r, ow, oh = ratio.view(B, 1).float(), dw.view(B, 1).float(), dh.view(B, 1).float()
    x1 = ((x - w / 2) - ow) / r; x2 = ((x + w / 2) - ow) / r
    y1 = ((y - h / 2) - oh) / r; y2 = ((y + h / 2) - oh) / r
    H, W = frame_hw[:, 0:1].float(), frame_hw[:, 1:2].float()
    boxes = torch.stack([x1.clamp(min=0), y1.clamp(min=0), torch.minimum(x2, W), torch.minimum(y2, H)], dim=2)
    return boxes, score, feats, torch.stack([sc, a, gj, gi], dim=2)


def temporal_fusion(center_feat: torch.Tensor, ref_feat: torch.Tensor, ref_score: torch.Tensor,
                    valid: Optional[torch.Tensor] = None):
    """Batched post_processing.py:246-278.

    center_feat (B,k,E)   candidates of the centre frame
    ref_feat    (B,R,k,E) candidates of each frame of the window (centre included, like the reference)
    ref_score   (B,R,k)   their confidences
    valid       (B,R) bool, False where a neighbour's cache was missing (its weight is zeroed *after*
                the softmax, :266-269)
    Returns (best (B,) int64 index of the winning centre candidate, fused (B,k))."""
    sim = torch.einsum("bce,brie->bcri", center_feat, ref_feat)               # (B,k_centre,R,k_ref)
    sim_max, sim_idx = sim.max(dim=3)                                          # best match in each frame  :258
    refer = torch.gather(ref_score.unsqueeze(1).expand(-1, sim.shape[1], -1, -1), 3, sim_idx.unsqueeze(3)).squeeze(3)
    w = F.softmax(sim_max, dim=2)                                              # over the R frames  :264
    if valid is not None:
        w = w * valid.unsqueeze(1).to(w.dtype)
    fused = (w * refer).sum(dim=2)                                             # :271
    return fused.argmax(dim=1), fused


# ---- the reference's cache files ---------------------------------------------------------------------
def cache_file(cache_dir: str, img_path: str, batch_idx: int) -> str:
    """``<cache_dir>/<video>/<frame>_<batch_idx>.pth`` (test_DCNet.py:636-645, post_processing.py:181-188)."""
    vid, img = img_path.split("/")[-2], img_path.split("/")[-1]
    return os.path.join(cache_dir, vid, img.split(".JPEG")[0] + "_" + str(batch_idx) + ".pth")


def save_cache_entry(path: str, boxes: torch.Tensor, scores: torch.Tensor, feats: torch.Tensor) -> None:
    """One clip's candidates in the reference's layout (test_DCNet.py:648-653): boxes (k,4), scores (k,),
    feats (k,E) -> {'pred_bbox_topk': (k,1,4), 'pred_score_topk': [float]*k, 'visu_feat': (k,1,E)}."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"pred_bbox_topk": boxes.detach().cpu().unsqueeze(1),
                "pred_score_topk": [float(v) for v in scores.detach().cpu()],
                "visu_feat": feats.detach().cpu().unsqueeze(1)}, path)


def load_cache_entry(path: str):
    d = torch.load(path, weights_only=False)
    return d["pred_bbox_topk"], torch.tensor(d["pred_score_topk"], dtype=torch.float), d["visu_feat"]


def fuse_from_cache(cache_dir: str, im_ids: Sequence[str], batch_idx: int, num_frame_k: int, device="cuda"):
    """One window of post_processing.py:212-284 from cache files: reads the centre entry and the
    ``num_frame_k`` window entries (file index ``batch_idx + offset``), substitutes the centre entry for a
    missing neighbour and zeroes its weight, fuses on ``device``.  Returns (box (1,4), index, fused)."""
    c = int(num_frame_k / 2)
    centre_path = cache_file(cache_dir, im_ids[c], batch_idx)
    boxes, _, feat = load_cache_entry(centre_path)
    rf, rs, valid = [], [], []
    for off, frm in zip(range(-c, c + 1), range(num_frame_k)):                 # :222-236
        p = cache_file(cache_dir, im_ids[frm], batch_idx + off)
        ok = os.path.exists(p)
        _, s, f = load_cache_entry(p if ok else centre_path)
        rf.append(f[:, 0]); rs.append(s); valid.append(ok)
    best, fused = temporal_fusion(feat[:, 0].unsqueeze(0).to(device), torch.stack(rf).unsqueeze(0).to(device),
                                  torch.stack(rs).unsqueeze(0).to(device),
                                  torch.tensor(valid, device=device).unsqueeze(0))
    i = int(best[0])
    return boxes[i], i, fused[0]


@torch.no_grad()
def cache_candidates(model, image, word_id, word_mask, n_frame: int, topk: int, size: int,
                     ratio: torch.Tensor, dw: torch.Tensor, dh: torch.Tensor):
    """``save_cache`` for a batch of clips (test_DCNet.py:558-643): n_frame eval forward, then the top-k
    candidates of every clip's centre frame.  (The reference passes ``topk`` where the model expects
    ``n_frame`` (:581) — its cache therefore only works with topk == frames per clip; here they are
    separate arguments.)  ``ratio, dw, dh``: (B,) letterbox meta of the centre frames."""
    model.eval()
    outbox, _, _, corr_feat, _ = model(image, word_id, word_mask, n_frame)
    hw = torch.tensor([letterbox_frame(size, float(r), float(a), float(b)) for r, a, b in
                       zip(ratio.tolist(), dw.tolist(), dh.tolist())], device=image.device)
    return topk_candidates(list(outbox), list(corr_feat), size, topk, ratio.to(image.device), dw.to(image.device),
                           dh.to(image.device), hw)
