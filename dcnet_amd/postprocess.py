"""Top-k candidate cache and temporal post-processing on the device.

The reference does this one clip at a time with host loops and ``.cpu().numpy()`` round trips
(test_DCNet.py:546-705 ``save_cache`` / ``get_topk_pred_bbox``; post_processing.py:181-284).  Here both
stages are batched and free of host synchronisation.  On CUDA tensors they run on the hand-written kernels of
``csrc/post.hip`` (``dcn_post_topk``: radix select + decode + un-letterbox + feature gather in one launch per
batch; ``dcn_post_fusion``: similarity, per-frame maximum, softmax over frames and fusion in one launch) —
no vendor sort, no BLAS, and no CPU path (CPU tensors raise).  The same computation as stock tensor ops lives in
``tests/post_torch.py`` as the tests' second opinion.

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


def letterbox_frame(size: int, ratio: float, dw: float, dh: float) -> Tuple[int, int]:
    """(height, width) of the un-letterboxed image, with the reference's rounding (test_DCNet.py:615-624)."""
    top, bottom = round(float(dh) - 0.1), size - round(float(dh) + 0.1)
    left, right = round(float(dw) - 0.1), size - round(float(dw) + 0.1)
    return round((bottom - top) / float(ratio)), round((right - left) / float(ratio))


def topk_candidates(outbox: Sequence[torch.Tensor], corr_feat: Sequence[torch.Tensor], size: int, topk: int,
                    ratio: torch.Tensor, dw: torch.Tensor, dh: torch.Tensor, frame_hw: torch.Tensor,
                    anchor_imsize: int = 416):
    """Batched ``save_cache`` core (test_DCNet.py:587-643, :662-705) on ``csrc/post.hip``.

    outbox[s]    (B,15,g,g) modulated head output of the n_frame model
    corr_feat[s] (B,E,g,g)  its correspondence features (any strides)
    ratio, dw, dh (B,)      letterbox meta;  frame_hw (B,2) = un-letterboxed (height, width)
    Returns boxes (B,k,4) xyxy in original-image pixels, scores (B,k), feats (B,k,E),
    cells (B,k,4) int64 = (scale, anchor, gj, gi).  Equal confidences resolve to the lowest flat index (the reference takes
    the first exact match of the value, :684)."""
    if outbox[0].is_cuda:
        from . import ops
        from .losses import scaled_anchors
        B = outbox[0].shape[0]
        dev = outbox[0].device
        ob = [o.reshape(B, 15, o.shape[-2], o.shape[-1]).float().contiguous() for o in outbox]
        cf = [c if c.dtype == torch.float32 else c.float() for c in corr_feat]
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).reshape(B).contiguous()
        return ops.post_topk(ob, cf, scaled_anchors(size, dev, anchor_imsize), size, topk, f32(ratio), f32(dw), f32(dh),
                             frame_hw.to(device=dev, dtype=torch.int64).contiguous())
    raise RuntimeError("dcnet_amd.postprocess.topk_candidates: HIP kernels only — move the tensors to the GPU (no CPU path; the "
                       "stock-torch restatement the tests compare with lives in tests/post_torch.py)")


def temporal_fusion(center_feat: torch.Tensor, ref_feat: torch.Tensor, ref_score: torch.Tensor,
                    valid: Optional[torch.Tensor] = None):
    """Batched post_processing.py:246-278 on ``csrc/post.hip``.

    center_feat (B,k,E)   candidates of the centre frame
    ref_feat    (B,R,k,E) candidates of each frame of the window (centre included, like the reference)
    ref_score   (B,R,k)   their confidences
    valid       (B,R) bool, False where a neighbour's cache was missing (its weight is zeroed *after*
                the softmax, :266-269)
    Returns (best (B,) int64 index of the winning centre candidate, fused (B,k))."""
    if center_feat.is_cuda:
        from . import ops
        return ops.post_fusion(center_feat.float().contiguous(), ref_feat.float().contiguous(), ref_score.float().contiguous(), valid)
    raise RuntimeError("dcnet_amd.postprocess.temporal_fusion: HIP kernels only — move the tensors to the GPU (no CPU path)")


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
