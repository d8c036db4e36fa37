"""CPU oracle for the DCNet dual-correspondence hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU fp32 restatement of
the reference's algorithm for the path named in BASELINE.json (north_star):
``grounding_model.forward`` of model/DCNet_model.py (train, T=2 pairs) and of
model/test_DCNet_model.py (inference, ``n_frame``), including the Darknet-53 /
YOLOv3 backbone of model/darknet.py.  Only ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
(``dcnet_amd``/``model``) never does.

Parity status: PINNED.  ``oracle/make_goldens.py`` imports the real reference
from /root/reference in the build container (recipe: SURVEY.md §8c), runs it on
seeded inputs/weights, checks this restatement against it (<=1e-5) and writes
the fixtures in tests/golden/ which tests/test_oracle_golden.py re-checks on
any machine.  The reference owns no tests or golden vectors of its own
(SURVEY.md §4), so those fixtures are the pin.

Everything is functional over a ``state_dict`` (name -> tensor) that uses the
reference's own key names, so the same dict can be loaded into
the reference model, into the product model, and evaluated here.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# --------------------------------------------------------------------------
# Darknet-53 / YOLOv3 graph  (model/yolov3.cfg:1-788, parsed by
# model/darknet.py:99-116; module index i == cfg block i+1 because the [net]
# block is popped, model/darknet.py:166)
# --------------------------------------------------------------------------

def darknet_defs() -> List[dict]:
    """The 107 module slots of model/yolov3.cfg, generated from the network's
    regular structure instead of the cfg text.  Each entry:
    type in {conv, yoloconv, shortcut, route, upsample, yolo}; conv entries carry
    filters/size/stride/bn/leaky.  Checked slot-by-slot against the parsed cfg in
    oracle/make_goldens.py."""
    d: List[dict] = []

    def conv(f, k, s=1, bn=True, leaky=True, kind="conv"):
        d.append(dict(type=kind, filters=f, size=k, stride=s, bn=bn, leaky=leaky))

    def res(f):  # 1x1 f/2, 3x3 f, shortcut -3     (yolov3.cfg:43-59 pattern)
        conv(f // 2, 1)
        conv(f, 3)
        d.append(dict(type="shortcut", frm=-3))

    conv(32, 3)                                   # cfg:25
    for f, n in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        conv(f, 3, 2)                             # downsample, cfg:35,65,115,286,461
        for _ in range(n):
            res(f)
    # neck + heads, cfg:551-788
    for si, (f, route_to) in enumerate(((512, None), (256, 61), (128, 36))):
        if route_to is not None:
            d.append(dict(type="route", layers=[-4]))        # cfg:618,705
            conv(f, 1)                                       # cfg:621,708
            d.append(dict(type="upsample"))                  # cfg:629,716
            d.append(dict(type="route", layers=[-1, route_to]))  # cfg:632,719
        conv(f, 1); conv(2 * f, 3); conv(f, 1); conv(2 * f, 3)
        conv(f, 1, kind="yoloconv")                          # cfg:583,669,756 (tap = its INPUT)
        conv(2 * f, 3)                                       # dead head, cfg:591,677,764
        conv(255, 1, bn=False, leaky=False)                  # dead head, cfg:599,685,772
        d.append(dict(type="yolo"))                          # cfg:607,693,780
    assert len(d) == 107
    return d


def _live_slots(defs: Sequence[dict]) -> List[bool]:
    """Slots whose output can reach a tap (F7: the three YOLO heads are dead)."""
    n = len(defs)
    need = [False] * n
    taps = [i for i, m in enumerate(defs) if m["type"] == "yoloconv"]
    stack = [t - 1 for t in taps]
    while stack:
        i = stack.pop()
        if i < 0 or need[i]:
            continue
        need[i] = True
        m = defs[i]
        if m["type"] in ("conv", "yoloconv", "upsample"):
            stack.append(i - 1)
        elif m["type"] == "shortcut":
            stack.append(i - 1); stack.append(i + m["frm"])
        elif m["type"] == "route":
            for l in m["layers"]:
                stack.append(l if l >= 0 else i + l)
    return need


def _bn(x: Tensor, sd: SD, p: str, training: bool, momentum: float, eps: float = 1e-5) -> Tensor:
    """nn.BatchNorm{1,2}d forward.  In training mode the running stats in ``sd``
    are updated in place exactly like the module does (and num_batches_tracked
    incremented), so the caller should hand in a clone if it wants to keep them."""
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, momentum, eps)


def darknet_forward(sd: SD, x: Tensor, training: bool, prefix: str = "visumodel.",
                    skip_dead: bool = True) -> List[Tensor]:
    """Darknet.forward, model/darknet.py:391-431.  Returns the three taps = the
    INPUT of each ``yoloconvolutional`` slot (:406-408): [1024@S/32, 512@S/16,
    256@S/8].  ``skip_dead`` skips the YOLO heads whose results the reference
    computes and throws away (F7) — except that in training mode their BatchNorm
    running stats would be updated by the reference, so pass skip_dead=False when
    comparing running-stat buffers of the dead heads."""
    defs = darknet_defs()
    live = _live_slots(defs)
    outs: List[Optional[Tensor]] = []
    taps: List[Tensor] = []
    for i, m in enumerate(defs):
        t = m["type"]
        if skip_dead and not live[i] and t != "yoloconv":
            outs.append(None); continue
        if t in ("conv", "yoloconv"):
            if t == "yoloconv":
                taps.append(x)                               # darknet.py:407
                if skip_dead and not live[i]:
                    outs.append(None); continue
            p = f"{prefix}module_list.{i}."
            k = m["size"]
            pad = (k - 1) // 2                               # darknet.py:176
            y = F.conv2d(x, sd[p + f"conv_{i}.weight"], sd.get(p + f"conv_{i}.bias"),
                         stride=m["stride"], padding=pad)    # darknet.py:179-186
            if m["bn"]:
                y = _bn(y, sd, p + f"batch_norm_{i}", training, 0.1)   # darknet.py:189
            if m["leaky"]:
                y = F.leaky_relu(y, 0.1)                     # darknet.py:191
            x = y
        elif t == "upsample":                                # MyUpsample2, darknet.py:158-160
            x = x[:, :, :, None, :, None].expand(-1, -1, -1, 2, -1, 2).reshape(
                x.size(0), x.size(1), x.size(2) * 2, x.size(3) * 2)
        elif t == "route":                                   # darknet.py:400-402
            x = torch.cat([outs[l if l >= 0 else i + l] for l in m["layers"]], 1)
        elif t == "shortcut":                                # darknet.py:403-405
            x = outs[-1] + outs[i + m["frm"]]
        elif t == "yolo":                                    # dead detection layer (:409-418)
            pass
        outs.append(x)
    return taps


# --------------------------------------------------------------------------
# Head building blocks
# --------------------------------------------------------------------------

def conv_bn_relu(sd: SD, p: str, x: Tensor, k: int, training: bool) -> Tensor:
    """ConvBatchNormReLU, model/darknet.py:118-156: conv (no bias) + BN2d(eps 1e-5,
    momentum 0.999) + ReLU (the scripts never pass leaky=True)."""
    y = F.conv2d(x, sd[p + ".conv.weight"], None, 1, (k - 1) // 2)
    y = _bn(y, sd, p + ".bn", training, 0.999)
    return F.relu(y)


def generate_coord(batch: int, height: int, width: int) -> Tensor:
    """model/DCNet_model.py:23-39 (without the .cuda()).  Note ``xv`` indexes
    rows although it is called x."""
    xv, yv = torch.meshgrid([torch.arange(0, height), torch.arange(0, width)], indexing="ij")
    xv_min = (xv.float() * 2 - width) / width
    yv_min = (yv.float() * 2 - height) / height
    xv_max = ((xv + 1).float() * 2 - width) / width
    yv_max = ((yv + 1).float() * 2 - height) / height
    xv_ctr = (xv_min + xv_max) / 2
    yv_ctr = (yv_min + yv_max) / 2
    hmap = torch.ones(height, width) * (1. / height)
    wmap = torch.ones(height, width) * (1. / width)
    coord = torch.stack([xv_min, yv_min, xv_max, yv_max, xv_ctr, yv_ctr, hmap, wmap], 0)
    return coord.unsqueeze(0).repeat(batch, 1, 1, 1)


def rnn_encoder(sd: SD, ids: Tensor, training: bool, p: str = "textmodel.",
                drop_p: float = 0.0) -> Tuple[Tensor, Tensor, Tensor]:
    """RNNEncoder.forward, model/DCNet_model.py:140-188.  Embedding -> Dropout ->
    Linear+ReLU -> packed BiLSTM (1 layer, hidden 512) -> unsort.  Returns
    (sentence (N,1024) = output at step len-1, context (N,L,1024), embedded (N,L,512)).
    Dropout (p=0.2 in the reference, :246) is disabled unless ``drop_p`` is set:
    the goldens are captured with p forced to 0."""
    lengths = (ids != 0).sum(1)                              # :150
    assert int(lengths.max()) == ids.size(1)                 # :158
    emb = F.embedding(ids, sd[p + "embedding.weight"])       # :168
    if training and drop_p > 0:
        emb = F.dropout(emb, drop_p, True)                   # :169
    emb = F.relu(F.linear(emb, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))  # :170
    N, L, _ = emb.shape
    H = sd[p + "rnn.weight_hh_l0"].shape[1]
    out = emb.new_zeros(N, L, 2 * H)
    # A packed-sequence LSTM == per-row LSTM over the first len_i steps (forward
    # direction) / over steps len_i-1..0 (reverse direction); padded steps are 0
    # (pad_packed_sequence, :182).  Written out as explicit cell recurrences.
    for d, sfx in enumerate(("", "_reverse")):
        w_ih, w_hh = sd[p + "rnn.weight_ih_l0" + sfx], sd[p + "rnn.weight_hh_l0" + sfx]
        b = sd[p + "rnn.bias_ih_l0" + sfx] + sd[p + "rnn.bias_hh_l0" + sfx]
        xg = F.linear(emb, w_ih, b)                          # (N,L,4H) gate order i,f,g,o
        h = emb.new_zeros(N, H); c = emb.new_zeros(N, H)
        steps = range(L) if d == 0 else range(L - 1, -1, -1)
        for t in steps:
            valid = (t < lengths).unsqueeze(1)
            g = xg[:, t] + F.linear(h, w_hh)
            i_, f_, g_, o_ = g.chunk(4, 1)
            c_new = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
            h_new = torch.sigmoid(o_) * torch.tanh(c_new)
            # rows past their length keep (h,c)=0 in the reverse direction until
            # their own last token, and emit zeros in either direction
            h = torch.where(valid, h_new, h)
            c = torch.where(valid, c_new, c)
            out[:, t, d * H:(d + 1) * H] = torch.where(valid, h_new, torch.zeros_like(h_new))
    emb = emb * (torch.arange(L).unsqueeze(0) < lengths.unsqueeze(1)).unsqueeze(2).to(emb.dtype)  # :178
    sent = out[torch.arange(N), lengths - 1]                 # :185-188
    return sent, out, emb


def phrase_attention(sd: SD, p: str, context: Tensor, embedded: Tensor, ids: Tensor
                     ) -> Tuple[Tensor, Tensor]:
    """PhraseAttention.forward, model/DCNet_model.py:196-219 (implicit softmax dim
    for a 2-D input is 1)."""
    s = F.linear(context, sd[p + ".fc.weight"], sd[p + ".fc.bias"]).squeeze(2)
    attn = F.softmax(s, dim=1)
    attn = attn * (ids != 0).float()
    attn = attn / attn.sum(1, keepdim=True)
    return attn, torch.bmm(attn.unsqueeze(1), embedded).squeeze(1)


def co_attention(f1: Tensor, f2: Tensor, temperature: float = 10.0) -> Tuple[Tensor, Tensor]:
    """Inter-frame co-attention for one scale, model/DCNet_model.py:449-459
    (== model/test_DCNet_model.py:259-274).  f1,f2: (b,C,H,W) unit-norm over C.
    Returns (f1_attn, f2_attn), both (b,C,H,W)."""
    b, c, h, w = f1.shape
    key = f1.reshape(b, c, h * w).transpose(1, 2).contiguous()       # :451
    value = f2.reshape(b, c, h * w).contiguous()                     # :452
    A = torch.bmm(key, value)                                        # :453  A[i,j]=<f1_i,f2_j>
    co2 = F.softmax(A.transpose(1, 2) * temperature, dim=1)          # :455
    co1 = F.softmax(A * temperature, dim=1)                          # :456
    f1_attn = torch.bmm(value, co2).view(b, c, h, w)                 # :458
    f2_attn = torch.bmm(key.transpose(1, 2), co1).view(b, c, h, w)   # :459
    return f1_attn, f2_attn


def interframe_sampling(f1: Tensor, f2: Tensor, top_k: int = 30, neg_n: int = 10, index: Optional[Tensor] = None):
    """Correspondence-patch sampling at scale 0, model/DCNet_model.py:381-430.
    f1,f2: (b,C,H,W).  Uses Python's global ``random`` in the reference's call
    order.  Returns (frame_feature, corrspendence_feature, neg_feature, idx) where
    the first three are lists of ``top_k`` tensors (b,C),(b,C),(b,neg_n,C) and idx
    = dict(q=(b,top_k), k=(b,top_k), neg=(b,top_k,neg_n), flat=(b,top_k), values=(b,top_k)).
    ``index`` (b,top_k) overrides the top-k selection: two fp32 implementations can order
    near-tied affinities differently, so parity tests feed the device's choice in here and
    check separately that it is a valid top-k up to rounding."""
    b, c, h, w = f1.shape
    hw = h * w
    p1, p2 = f1.flatten(-2), f2.flatten(-2)
    cmap = torch.bmm(p1.permute(0, 2, 1), p2).flatten(-2)            # :390
    qi = torch.empty(b, top_k, dtype=torch.long)
    ki = torch.empty(b, top_k, dtype=torch.long)
    ni = torch.empty(b, top_k, neg_n, dtype=torch.long)
    flat = torch.empty(b, top_k, dtype=torch.long)
    vals = torch.empty(b, top_k)
    for ii in range(b):
        v_, index_ = cmap[ii].topk(top_k, dim=0, largest=True, sorted=True)   # :395
        if index is not None:
            index_ = index[ii]
        flat[ii] = index_; vals[ii] = v_.detach()
        for jj in range(top_k):
            qi[ii, jj] = index_[jj] // hw                            # :407
            kk = int(index_[jj] % hw)                                # :409
            ki[ii, jj] = kk
            lst = list(range(hw)); lst.remove(kk)                    # :411-412
            ni[ii, jj] = torch.tensor(random.sample(lst, neg_n))     # :413
    ar = torch.arange(b)
    frame = [p1[ar, :, qi[:, j]] for j in range(top_k)]
    corr = [p2[ar, :, ki[:, j]] for j in range(top_k)]
    neg = [p2.permute(0, 2, 1)[ar.unsqueeze(1), ni[:, j]] for j in range(top_k)]
    return frame, corr, neg, dict(q=qi, k=ki, neg=ni, flat=flat, values=vals, cmap=cmap.detach())


def crossmodal_correspondence(lag: Tensor, vit: Tensor, lv_map: Tensor, neg_n: int = 5, cols: Optional[Tensor] = None):
    """Crossmodal_corrspondence, model/DCNet_model.py:41-112 (top_k=1).
    lag (N,L,E) vit (N,E,HW0) lv_map (N,L,HW0).  The ``index`` loop (:81-90) draws
    N samples per (ii,jj) but only the last one (index=N-1) is kept (:94)."""
    lv = lv_map.permute(0, 2, 1); v = vit.permute(0, 2, 1); l = lag.permute(0, 2, 1)   # :43-45
    N, rows = v.shape[0], v.shape[1]
    if cols is None:
        cols = lv.topk(1, dim=2, largest=True, sorted=True)[1][..., 0]   # (N,rows)   :48
    ni = torch.empty(N, rows, neg_n, dtype=torch.long)
    for ii in range(N):
        for jj in range(rows):
            for index in range(N):
                lst = list(range(rows))
                if index == ii:
                    lst.remove(jj)                                   # :83-84
                s = random.sample(lst, neg_n)                        # :87
            ni[ii, jj] = torch.tensor(s)
    ar = torch.arange(N)
    vit_pos = [v[:, j, :] for j in range(rows)]                              # (N,E)
    lag_pos = [l[ar, :, cols[:, j]].unsqueeze(1) for j in range(rows)]       # (N,1,E)
    neg = [v[N - 1][ni[:, j]] for j in range(rows)]                          # (N,neg_n,E) from image N-1
    return vit_pos, lag_pos, neg, dict(word=cols, neg=ni, lv=lv.detach())


# --------------------------------------------------------------------------
# Head shared by the train (pairs) and test (n_frame) models
# --------------------------------------------------------------------------

def _head(sd: SD, corr_feat: List[Tensor], word_id: Tensor, training: bool,
          drop_p: float = 0.0):
    """Language + fusion + scores + conf modulation on B rows:
    model/DCNet_model.py:471-621 == model/test_DCNet_model.py:339-477."""
    B = corr_feat[0].shape[0]
    max_len = int((word_id != 0).sum(1).max())                       # :474
    word_id = word_id[:, :max_len]                                   # :475
    raw_flang, context, embedded = rnn_encoder(sd, word_id, training, drop_p=0.2 if drop_p else 0.0)
    # mapping_lang, :268-276,485
    x = F.linear(raw_flang, sd["mapping_lang.0.weight"], sd["mapping_lang.0.bias"])
    x = F.relu(_bn(x, sd, "mapping_lang.1", training, 0.1))
    if training and drop_p > 0:
        x = F.dropout(x, drop_p, True)
    x = F.linear(x, sd["mapping_lang.4.weight"], sd["mapping_lang.4.bias"])
    x = F.relu(_bn(x, sd, "mapping_lang.5", training, 0.1))
    flang = F.normalize(x, p=2, dim=1)                               # :487

    coord_list, outbox = [], []
    for ii in range(3):
        h, w = corr_feat[ii].shape[2:]
        tile = flang.view(B, -1, 1, 1).repeat(1, 1, h, w)            # :492-493
        coord = generate_coord(B, h, w); coord_list.append(coord)    # :495-496
        z = torch.cat([corr_feat[ii], tile, coord], dim=1)           # :497
        j = 0
        while f"fcn_emb.{ii}.{j}.conv.weight" in sd:                 # fcn_emb :314-327 (three blocks; light=True: one, :296-303)
            z = conv_bn_relu(sd, f"fcn_emb.{ii}.{j}", z, sd[f"fcn_emb.{ii}.{j}.conv.weight"].shape[-1], training)
            j += 1
        if f"fcn_out.{ii}.0.conv.weight" in sd:                      # :328-338
            z = conv_bn_relu(sd, f"fcn_out.{ii}.0", z, 1, training)
            z = F.conv2d(z, sd[f"fcn_out.{ii}.1.weight"], sd[f"fcn_out.{ii}.1.bias"])
        else:                                                        # light=True: a bare Conv2d (:304-312)
            z = F.conv2d(z, sd[f"fcn_out.{ii}.0.weight"], sd[f"fcn_out.{ii}.0.bias"])
        outbox.append(z)

    _, flang_attn = phrase_attention(sd, "sub_attn", context, embedded, word_id)   # :525
    flang_attn = F.normalize(flang_attn, p=2, dim=1).unsqueeze(2).unsqueeze(2)     # :526-528
    sim_score = [torch.sum(flang_attn * corr_feat[ii], dim=1) for ii in range(3)]  # :530-535

    obj_score, only_obj = [], []
    for ii in range(3):
        b, _, h, w = outbox[ii].shape
        conf = outbox[ii].view(b, 3, 5, h, w)[:, :, 4].mean(dim=1)   # :548-551
        obj_score.append(conf * sim_score[ii]); only_obj.append(conf)

    _, flang_loc = phrase_attention(sd, "loc_attn", context, embedded, word_id)    # :556
    flang_loc = F.normalize(flang_loc, p=2, dim=1)                                 # :557

    coord_map = torch.cat([c.reshape(B, 8, -1).permute(0, 2, 1) for c in coord_list], dim=1)  # :565-567
    obj_map = torch.cat([o.reshape(B, -1) for o in obj_score], dim=1)              # :566-568
    obj_map = F.normalize(obj_map, p=2, dim=1)                                     # :569
    P = obj_map.shape[1]
    ce = F.linear(coord_map.reshape(-1, 8), sd["loc_embedding.0.weight"], sd["loc_embedding.0.bias"])
    ce = F.relu(_bn(ce, sd, "loc_embedding.1", training, 0.1)).view(B, P, 8)       # :573-575
    ce = F.normalize(ce, p=2, dim=2)                                               # :578
    rel = torch.bmm(ce, ce.permute(0, 2, 1)) * obj_map.unsqueeze(1)                # :581-582
    rel = F.linear(rel.reshape(-1, P), sd["loc_text_embedding.0.weight"],
                   sd["loc_text_embedding.0.bias"])                                # :584-585 (1344 -> P)
    rel = F.relu(_bn(rel, sd, "loc_text_embedding.1", training, 0.1))
    rel = F.normalize(rel.view(B, P, -1).permute(0, 2, 1), p=2, dim=1)             # :587-589
    loc_map = torch.sum(rel * flang_loc.unsqueeze(-1), dim=1)                      # :593-594
    mn = loc_map.min(dim=1)[0].unsqueeze(1); mx = loc_map.max(dim=1)[0].unsqueeze(1)
    loc_map = (loc_map - mn) / (mx - mn + 1e-6)                                    # :597
    loc_score, s = [], 0
    for ii in range(3):
        h, w = corr_feat[ii].shape[2:]
        loc_score.append(loc_map[:, s:s + h * w].reshape(-1, h, w)); s += h * w    # :604-610

    final = []
    for ii in range(3):                                                            # :612-621
        b, _, h, w = outbox[ii].shape
        ob = outbox[ii].view(b, 3, 5, h, w)
        conf = ob[:, :, 4] * sim_score[ii].unsqueeze(1) * loc_score[ii].unsqueeze(1)
        ob = torch.cat([ob[:, :, :4], conf.unsqueeze(2)], dim=2)
        final.append(ob.view(b, 15, h, w))
    return final, sim_score, loc_score, only_obj, flang_attn, context


def _map_norm(sd: SD, raw: List[Tensor], training: bool) -> List[Tensor]:
    """mapping_visu + L2-norm over C, model/DCNet_model.py:356-359."""
    return [F.normalize(conv_bn_relu(sd, f"mapping_visu.{i}", raw[i], 1, training), p=2, dim=1)
            for i in range(3)]


def grounding_forward_pairs(sd: SD, image: Tensor, word_id: Tensor, training: bool,
                            sample: bool = True, drop_p: float = 0.0, skip_dead: bool = True,
                            k9_index: Optional[Tensor] = None, k14_cols: Optional[Tensor] = None) -> dict:
    """grounding_model.forward of model/DCNet_model.py:340-650 (T=2 pair
    semantics, F1).  Returns a dict with every tensor of the 11-tuple (train) /
    4-tuple (eval) plus a few intermediates used by the parity tests."""
    N = image.shape[0]
    assert N % 2 == 0
    raw = darknet_forward(sd, image, training, skip_dead=skip_dead)              # :344
    fv = _map_norm(sd, raw, training)
    pairs = [f.view(N // 2, 2, *f.shape[1:]) for f in fv]                        # :365-367
    in1 = [p[:, 0] for p in pairs]; in2 = [p[:, 1] for p in pairs]               # :370-374
    res = dict(taps=raw, fvisu=fv)
    if sample:
        fr, co, ng, idx = interframe_sampling(in1[0], in2[0], index=k9_index)    # :381-430
        res.update(frame_feature=fr, corrspendence_feature=co, neg_feature=ng, k9_idx=idx)
    corr = []
    for ii in range(3):                                                          # :449-464
        a1, a2 = co_attention(in1[ii], in2[ii])
        c1 = torch.cat([in1[ii], a1], 1).unsqueeze(1)
        c2 = torch.cat([in2[ii], a2], 1).unsqueeze(1)
        corr.append(torch.cat([c1, c2], dim=1).reshape(N, -1, *a1.shape[2:]))
    res["coattn_cat"] = corr
    corr = [F.normalize(conv_bn_relu(sd, f"corr_conv.{ii}.0", corr[ii], 1, training), p=2, dim=1)
            for ii in range(3)]                                                  # :467-469
    outbox, sim, loc, only_obj, flang_attn, context = _head(sd, corr, word_id, training, drop_p)
    res.update(outbox=outbox, sim_score=sim, loc_score=loc, only_obj=only_obj,
               corr_feat=corr, flang_attn=flang_attn)
    if sample:                                                                   # :625-637
        vit = F.normalize(fv[0].flatten(-2), dim=2)          # over positions (!)  :629
        lag = F.normalize(context[:, :, 0::2], dim=1)        # interpolate(0.5)=ch 0,2,4..; over L  :631-632
        lv = torch.bmm(lag, vit)                                                 # :634
        lv = F.conv1d(lv, sd["feature_map.0.weight"], sd["feature_map.0.bias"], padding=1)
        lv = F.softmax(lv, dim=1)                                                # :287-290,635
        vp, lp, nc, idx = crossmodal_correspondence(lag, vit, lv, cols=k14_cols) # :637
        res.update(vit_posit=vp, lag_posit=lp, neg_cross=nc, k14_idx=idx)
    return res


def grounding_forward_nframe(sd: SD, image: Tensor, word_id: Tensor, n_frame: int,
                             training: bool = False, drop_p: float = 0.0) -> dict:
    """grounding_model.forward of model/test_DCNet_model.py:284-483 (centre frame
    co-attends to each other frame, F2)."""
    B = image.shape[0] // n_frame                                                # :287
    raw = darknet_forward(sd, image, training)
    fv = _map_norm(sd, raw, training)
    clips = [f.view(B, n_frame, *f.shape[1:]) for f in fv]                       # :299-301
    ctr = n_frame // 2                                                           # :303
    in1 = [c[:, ctr] for c in clips]
    sets = []
    for idx in range(n_frame):                                                   # :312-320
        if idx == ctr:
            continue
        cf = []
        for ii in range(3):                                                      # cal_corr_feat :247-282
            a1, _ = co_attention(in1[ii], clips[ii][:, idx])
            z = conv_bn_relu(sd, f"corr_conv.{ii}.0", torch.cat([in1[ii], a1], 1), 1, training)
            cf.append(F.normalize(z, p=2, dim=1))
        sets.append(cf)
    corr = [torch.stack([s[ii] for s in sets], 0).mean(0) for ii in range(3)]    # :324-332
    outbox, sim, loc, only_obj, flang_attn, _ = _head(sd, corr, word_id, training, drop_p)
    return dict(taps=raw, fvisu=fv, outbox=outbox, sim_score=sim, loc_score=loc,
                corr_feat=corr, only_obj=only_obj, flang_attn=flang_attn)


# --------------------------------------------------------------------------
# Caller-side pieces needed to drive backward / report Acc@0.5
# (train_DCNet.py; the "next" rows of SURVEY.md §8f, restated for the tests)
# --------------------------------------------------------------------------

ANCHORS_FULL = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119),
                (116, 90), (156, 198), (373, 326)][::-1]      # train_DCNet.py:404-406


def decode_boxes(outbox: List[Tensor], size: int, anchor_imsize: int = 416) -> Tensor:
    """Eval decode, train_DCNet.py:764-810: global argmax of conf over 3 scales x 3
    anchors, box = (sigmoid(tx)+gi, sigmoid(ty)+gj, exp(tw)*aw, exp(th)*ah)*stride,
    xywh -> xyxy (utils/utils.py:34-40)."""
    N = outbox[0].shape[0]
    ob = [o.view(N, 3, 5, o.shape[2], o.shape[3]) for o in outbox]
    conf = torch.cat([o[:, :, 4].reshape(N, -1) for o in ob], dim=1)
    _, loc = conf.max(dim=1)
    out = torch.zeros(N, 4)
    for ii in range(N):
        l = int(loc[ii]); sc = 0; base = 0
        for sc in range(3):
            g = size // (32 // 2 ** sc)
            if l < base + 3 * g * g:
                break
            base += 3 * g * g
        g, gs = size // (32 // 2 ** sc), 32 // 2 ** sc
        l -= base
        n, gj, gi = l // (g * g), (l % (g * g)) // g, l % g
        aw, ah = [(a[0] / (anchor_imsize / g), a[1] / (anchor_imsize / g))
                  for a in ANCHORS_FULL[3 * sc:3 * sc + 3]][n]
        t = ob[sc][ii, n, :, gj, gi]
        x = (torch.sigmoid(t[0]) + gi) * gs; y = (torch.sigmoid(t[1]) + gj) * gs
        w = torch.exp(t[2]) * aw * gs; h = torch.exp(t[3]) * ah * gs
        out[ii] = torch.stack([x - w / 2, y - h / 2, x + w / 2, y + h / 2])
    return out


def bbox_iou_xyxy(b1: Tensor, b2: Tensor) -> Tensor:
    """utils/utils.py:76-104 with x1y1x2y2=True (no +1 pixel convention)."""
    ix1 = torch.max(b1[:, 0], b2[:, 0]); iy1 = torch.max(b1[:, 1], b2[:, 1])
    ix2 = torch.min(b1[:, 2], b2[:, 2]); iy2 = torch.min(b1[:, 3], b2[:, 3])
    inter = torch.clamp(ix2 - ix1, 0) * torch.clamp(iy2 - iy1, 0)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    return inter / (a1 + a2 - inter + 1e-16)
