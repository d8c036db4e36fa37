"""Stock-torch restatement of dcnet_amd.postprocess (top-k candidates, temporal fusion): the tests' second opinion beside the
reference-generated fixtures.  Test infrastructure — the product (csrc/post.hip) never imports it."""
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from dcnet_amd.losses import ANCHORS_FULL, _const


def topk_candidates_torch(outbox: Sequence[torch.Tensor], corr_feat: Sequence[torch.Tensor], size: int, topk: int,
                          ratio: torch.Tensor, dw: torch.Tensor, dh: torch.Tensor, frame_hw: torch.Tensor,
                          anchor_imsize: int = 416):
    """Batched ``save_cache`` core (test_DCNet.py:587-643, :662-705) as stock tensor ops.

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



def temporal_fusion_torch(center_feat: torch.Tensor, ref_feat: torch.Tensor, ref_score: torch.Tensor,
                          valid: Optional[torch.Tensor] = None):
    """Batched post_processing.py:246-278 as stock tensor ops.

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


