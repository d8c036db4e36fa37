"""Target assignment, the five training losses and the evaluation decode of DCNet on the device.

The reference computes these on the caller side (train_DCNet.py:45-220 losses, :265-332 build_target, :613-642
combination, :764-816 decode) with per-sample Python loops and ``.item()`` host syncs.  Here every one of them is a HIP
kernel of libdcnet_hip.so (csrc/loss.hip) behind an autograd node (functions.DenseLosses / functions.Contrastive), so one
training step issues no host synchronisation between forward and backward.  There is no CPU path: tensors must live on
the GPU.  The CPU restatement the tests compare against is oracle/train_oracle.py.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from . import ops
from .functions import Contrastive, DenseLosses, RowDot

ANCHORS_FULL = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119),
                (116, 90), (156, 198), (373, 326)][::-1]          # train_DCNet.py:404-406 (reversed)

_const_cache = {}


def _const(key, device, builder):
    """Small constant tensors, uploaded once per (key, device): a ``torch.tensor(list, device=...)`` in the
    step would be a pageable H2D copy that blocks the host until the queued forward kernels have drained."""
    k = (key, str(device))
    t = _const_cache.get(k)
    if t is None:
        t = builder().to(device)
        _const_cache[k] = t
    return t


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"dcnet_amd.losses.{what}: HIP kernels only — move the tensors to the GPU (the CPU restatement "
                           "used by the tests lives in oracle/train_oracle.py)")


def scaled_anchors(size: int, device, anchor_imsize: int = 416) -> torch.Tensor:
    """(3,3,2) float32: anchors of scale s divided by (anchor_imsize / grid_s), computed in double and rounded once, as the
    reference's Python does (train_DCNet.py:293-296,789-793).  Uploaded once per (size, device): a tensor built from a list
    inside the step would be a pageable H2D copy that blocks the host until the queued forward kernels have drained."""
    key = ("anchors", size, anchor_imsize, str(device))
    t = _const_cache.get(key)
    if t is None:
        rows = []
        for s in range(3):
            g = size // (32 // (2 ** s))
            rows.append([(a[0] / (anchor_imsize / g), a[1] / (anchor_imsize / g)) for a in ANCHORS_FULL[3 * s:3 * s + 3]])
        t = torch.tensor(rows, dtype=torch.float64).to(torch.float32).to(device)
        _const_cache[key] = t
    return t


def compact_target(raw_coord: torch.Tensor, size: int, anchor_imsize: int = 416):
    """build_target (train_DCNet.py:265-332) in the form the loss kernels read: target_i (N,4) int32 = best anchor (0..8), gi,
    gj, flat cell index over the scale-concatenated positions; target_f (N,4) = tx, ty, tw, th.  Boxes are clamped to
    [0, size-1] (:605)."""
    _need_gpu(raw_coord, "compact_target")
    return ops.build_target(raw_coord.contiguous().float(), scaled_anchors(size, raw_coord.device, anchor_imsize), size)


def build_target(raw_coord: torch.Tensor, size: int, anchor_imsize: int = 416):
    """train_DCNet.py:265-332 with the reference's return values: (bbox_list[3] (N,3,5,g,g), gi (N,), gj (N,), best_n (N,),
    bbox_center_list[3] (N,5,g,g))."""
    ti, tf = compact_target(raw_coord, size, anchor_imsize)
    box, ctr = ops.target_dense(ti, tf, size)
    return box, ti[:, 1].long(), ti[:, 2].long(), ti[:, 0].long(), ctr


def _stacked(lst) -> torch.Tensor:
    """The (n, K, ...) tensor whose unbind(1) is ``lst`` — carried by the model's OutputList, rebuilt (a copy) for plain lists."""
    t = getattr(lst, "stacked", None)
    return t if t is not None else torch.stack(list(lst), dim=1)


def interframe_contrastive_loss(q_list, k_list, neg_list, T: float = 0.07):
    """train_DCNet.py:114-136: lists over the top-k matches of (n,c), (n,c), (n,m,c) tensors -> mean InfoNCE over all rows."""
    q, k, neg = _stacked(q_list), _stacked(k_list), _stacked(neg_list)
    _need_gpu(q, "interframe_contrastive_loss")
    return Contrastive.apply(q, k, neg, T)


def crossmodal_contrastive_loss(q_list, k_list, neg_list, T: float = 0.07):
    """train_DCNet.py:140-166 with one positive word per position (Crossmodal_corrspondence is called with top_k = 1,
    model/DCNet_model.py:637, so k is (n,1,c))."""
    q, k, neg = _stacked(q_list), _stacked(k_list), _stacked(neg_list)
    _need_gpu(q, "crossmodal_contrastive_loss")
    if k.dim() == 4:
        if k.shape[2] != 1:
            raise NotImplementedError("crossmodal_contrastive_loss: one positive per row (the model samples top_k = 1)")
        k = k.squeeze(2)
    return Contrastive.apply(q, k, neg, T)


def neg_sim_score(corr_feat: Sequence[torch.Tensor], flang_attn: torch.Tensor) -> List[torch.Tensor]:
    """train_DCNet.py:623-627: <flang_attn reversed along the batch, corr_feat[:, :512]> per position, for outputs that do
    not carry the model's fused ``neg_sim`` (plain lists).  corr_feat[s] is logically NCHW."""
    n = flang_attn.shape[0]
    q = flang_attn.reshape(n, -1)
    return [RowDot.apply(cf[:, :q.shape[1]].permute(0, 2, 3, 1), q, True) for cf in corr_feat]


def dense_losses(pred, sim, neg_sim, loc, bbox: torch.Tensor, size: int):
    """(yolo_loss :45-72, rank_loss :173-203, loc_loss :205-220) for predictions pred[s] (N,15,g,g) or (N,3,5,g,g)."""
    _need_gpu(pred[0], "dense_losses")
    ti, tf = compact_target(bbox, size)
    n = pred[0].shape[0]
    p15 = [p.reshape(n, 15, p.shape[-2], p.shape[-1]) for p in pred]
    return DenseLosses.apply(*p15, *sim, *neg_sim, *loc, ti, tf, size)


def total_loss(outputs, bbox: torch.Tensor, size: int):
    """train_DCNet.py:605-642 on the 11-tuple of grounding_model.forward (train mode).
    Returns (loss, dict of the five parts)."""
    (pred, sim, loc, corr_feat, flang_attn, frame_f, corr_f, neg_f, vit_p, lag_p, neg_c) = outputs
    _need_gpu(pred[0], "total_loss")
    neg_sim = getattr(sim, "neg_sim", None)
    if neg_sim is None:
        neg_sim = neg_sim_score(corr_feat, flang_attn)
    cs = getattr(frame_f, "stream", None)
    if cs is not None:
        # the sampled features were produced on the model's sampling stream: evaluate their two losses there as well, so
        # that autograd replays this whole branch beside the heads' backward
        main = torch.cuda.current_stream()
        with torch.cuda.stream(cs):
            inter = interframe_contrastive_loss(frame_f, corr_f, neg_f)
            cross = crossmodal_contrastive_loss(vit_p, lag_p, neg_c)
        main.wait_stream(cs)
        inter.record_stream(main); cross.record_stream(main)
    else:
        inter = interframe_contrastive_loss(frame_f, corr_f, neg_f)
        cross = crossmodal_contrastive_loss(vit_p, lag_p, neg_c)
    yolo, rank, locl = dense_losses(pred, sim, neg_sim, loc, bbox, size)
    parts = dict(yolo=yolo, rank=rank, interframe=inter, cross=cross, loc=locl)
    loss = yolo + 100 * rank + locl + 100 * inter + cross                                    # :642
    return loss, parts


def decode_boxes(outbox: Sequence[torch.Tensor], size: int, anchor_imsize: int = 416) -> torch.Tensor:
    """Evaluation decode (train_DCNet.py:764-810): global arg-max of the modulated confidence over 3 scales x 3 anchors, then
    (sigmoid(tx)+gi, sigmoid(ty)+gj, exp(tw)*aw, exp(th)*ah)*stride and xywh -> xyxy.  outbox[s] (N,15,g,g) or (N,3,5,g,g)."""
    _need_gpu(outbox[0], "decode_boxes")
    n = outbox[0].shape[0]
    ob = [o.reshape(n, 15, o.shape[-2], o.shape[-1]).contiguous().float() for o in outbox]
    return ops.decode_boxes(ob, scaled_anchors(size, ob[0].device, anchor_imsize), size)


def bbox_iou(box1: torch.Tensor, box2: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:76-104 (x1y1x2y2), row by row."""
    _need_gpu(box1, "bbox_iou")
    return ops.box_iou(box1.contiguous().float(), box2.contiguous().float())
