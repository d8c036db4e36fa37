"""Op-level parity of the round-2 kernels — phrase attention, the cross-scale head tail, the two correspondence-sampling
heads, the InfoNCE rows — against fp64 torch restatements of the reference lines they replace (model/DCNet_model.py:41-112,
190-219,381-430,545-637; train_DCNet.py:114-166), forward and backward."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _close(a, b, tol, name=""):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, (name, err, scale)


# ---- phrase attention ---------------------------------------------------------------------------------------------
def _ref_phrase(context, embedded, ids, w, b, normalize):
    att = F.softmax(F.linear(context, w, b).squeeze(2), dim=1)                 # :206-207
    att = att * (ids != 0).double()                                            # :210
    att = att / att.sum(1, keepdim=True)                                       # :211-212
    v = torch.bmm(att.unsqueeze(1), embedded).squeeze(1)                       # :215-216
    return att, (F.normalize(v, p=2, dim=1) if normalize else v)


@pytest.mark.parametrize("n,L,heads,normalize", [(6, 20, 2, True), (3, 20, 1, False), (5, 12, 2, True)])
def test_phrase_attention_fwd_bwd(dev, n, L, heads, normalize):
    from dcnet_amd.functions import PhraseAttn
    g = _g(n + L)
    D, E = 1024, 512
    ctx = torch.randn(n, L, D, generator=g); emb = torch.randn(n, L, E, generator=g)
    ids = torch.randint(1, 1000, (n, L), generator=g)
    ids[1, L - 5:] = 0; ids[2, L - 1:] = 0                                     # ragged queries
    ws = [torch.randn(1, D, generator=g) * 0.05 for _ in range(heads)]; bs = [torch.randn(1, generator=g) for _ in range(heads)]
    go = [torch.randn(n, E, generator=g) for _ in range(heads)]
    cd, ed = ctx.double().requires_grad_(True), emb.double().requires_grad_(True)
    wd = [w.double().requires_grad_(True) for w in ws]; bd = [b.double().requires_grad_(True) for b in bs]
    refs = [_ref_phrase(cd, ed, ids, wd[h], bd[h], normalize) for h in range(heads)]
    sum((r[1] * go[h].double()).sum() for h, r in enumerate(refs)).backward()
    cg, eg = ctx.to(dev).requires_grad_(True), emb.to(dev).requires_grad_(True)
    wg = [w.to(dev).requires_grad_(True) for w in ws]; bg = [b.to(dev).requires_grad_(True) for b in bs]
    o0, o1, attn = PhraseAttn.apply(cg, eg, ids.to(dev), wg[0], bg[0], wg[1] if heads == 2 else None, bg[1] if heads == 2 else None, normalize)
    outs = [o0, o1][:heads]
    sum((o * go[h].to(dev)).sum() for h, o in enumerate(outs)).backward()
    for h in range(heads):
        _close(attn[h], refs[h][0], 2e-6, "attn"); _close(outs[h], refs[h][1], 3e-6, "out")
        _close(wg[h].grad, wd[h].grad, 2e-5, "dw"); _close(bg[h].grad, bd[h].grad, 2e-5, "db")
    _close(cg.grad, cd.grad, 2e-5, "dcontext"); _close(eg.grad, ed.grad, 2e-5, "dembedded")


# ---- head tail ------------------------------------------------------------------------------------------------------
def _coord_rows(grids):
    from dcnet_amd.model import generate_coord_nhwc
    return torch.cat([generate_coord_nhwc(g, g, "cpu").reshape(-1, 8) for g in grids], 0)


def _ref_head_tail(logits, sim, q_loc, coord, P_, bn_state, training):
    """model/DCNet_model.py:545-621 as written (P x P relation tensor), fp64.  logits[s] (B,15,h,w) NCHW."""
    B = logits[0].shape[0]
    only, obj = [], []
    for s in range(3):
        b, _, h, w = logits[s].shape
        c = logits[s].view(b, 3, 5, h, w)[:, :, 4].mean(dim=1)                 # :551
        only.append(c); obj.append(c * sim[s])                                 # :550
    obj_map = F.normalize(torch.cat([o.reshape(B, -1) for o in obj], 1), p=2, dim=1)     # :566-569
    P = obj_map.shape[1]
    cm = coord.unsqueeze(0).repeat(B, 1, 1).reshape(-1, 8)                      # :572
    ce = F.linear(cm, P_["w_le"], P_["b_le"])
    ce = F.batch_norm(ce, bn_state["rm_le"], bn_state["rv_le"], P_["g_le"], P_["be_le"], training, 0.1, 1e-5)
    ce = F.normalize(F.relu(ce).view(B, P, 8), p=2, dim=2)                      # :573-578
    rel = torch.bmm(ce, ce.permute(0, 2, 1)) * obj_map.unsqueeze(1)             # :581-582
    rel = F.linear(rel.reshape(-1, P), P_["w_lt"], P_["b_lt"])                  # :584-585
    rel = F.relu(F.batch_norm(rel, bn_state["rm_lt"], bn_state["rv_lt"], P_["g_lt"], P_["be_lt"], training, 0.1, 1e-5))
    rel = F.normalize(rel.view(B, P, -1).permute(0, 2, 1), p=2, dim=1)          # :587-589
    loc_map = torch.sum(rel * q_loc.unsqueeze(-1), dim=1)                       # :593-594
    mn = loc_map.min(dim=1)[0].unsqueeze(1); mx = loc_map.max(dim=1)[0].unsqueeze(1)
    loc_map = (loc_map - mn) / (mx - mn + 1e-6)                                 # :597
    loc, st, outbox = [], 0, []
    for s in range(3):
        b, _, h, w = logits[s].shape
        loc.append(loc_map[:, st:st + h * w].reshape(b, h, w)); st += h * w
        ob = logits[s].view(b, 3, 5, h, w)
        conf = ob[:, :, 4] * sim[s].unsqueeze(1) * loc[s].unsqueeze(1)          # :612-621
        outbox.append(torch.cat([ob[:, :, :4], conf.unsqueeze(2)], 2).view(b, 15, h, w))
    return outbox, loc, only


@pytest.mark.parametrize("B,grids,training", [(3, (4, 8, 16), True), (2, (5, 10, 20), False), (4, (3, 6, 12), True)])
def test_head_tail_matches_the_reference_form(dev, B, grids, training):
    from dcnet_amd.functions import HeadTail
    g = _g(B * 7 + grids[0])
    P = sum(x * x for x in grids)
    coord = _coord_rows(grids)
    prm = dict(w_le=torch.randn(8, 8, generator=g) * 0.6, b_le=torch.randn(8, generator=g) * 0.1, g_le=1 + 0.2 * torch.randn(8, generator=g),
               be_le=0.3 * torch.randn(8, generator=g), w_lt=torch.randn(512, P, generator=g) / P ** 0.5, b_lt=torch.randn(512, generator=g) * 0.1,
               g_lt=1 + 0.2 * torch.randn(512, generator=g), be_lt=0.2 * torch.randn(512, generator=g))
    bn_le = torch.nn.BatchNorm1d(8); bn_lt = torch.nn.BatchNorm1d(512)
    with torch.no_grad():
        bn_le.running_mean.copy_(torch.randn(8, generator=g) * 0.2); bn_le.running_var.copy_(torch.rand(8, generator=g) + 0.5)
        bn_lt.running_mean.copy_(torch.randn(512, generator=g) * 0.02); bn_lt.running_var.copy_(torch.rand(512, generator=g) * 0.01 + 0.005)
    logits = [torch.randn(B, 15, x, x, generator=g) for x in grids]
    sim = [torch.randn(B, x, x, generator=g) * 0.5 for x in grids]
    q_loc = F.normalize(torch.randn(B, 512, generator=g), dim=1)
    g_out = [torch.randn(B, 15, x, x, generator=g) for x in grids]; g_loc = [torch.randn(B, x, x, generator=g) for x in grids]
    g_only = [torch.randn(B, x, x, generator=g) for x in grids]
    # ---- fp64 reference ----
    ld = [t.double().requires_grad_(True) for t in logits]; sd = [t.double().requires_grad_(True) for t in sim]
    qd = q_loc.double().requires_grad_(True); pd = {k: v.double().requires_grad_(True) for k, v in prm.items()}
    st = dict(rm_le=bn_le.running_mean.double().clone(), rv_le=bn_le.running_var.double().clone(),
              rm_lt=bn_lt.running_mean.double().clone(), rv_lt=bn_lt.running_var.double().clone())
    ro, rl, rn = _ref_head_tail(ld, sd, qd, coord.double(), pd, st, training)
    (sum((a * b.double()).sum() for a, b in zip(ro, g_out)) + sum((a * b.double()).sum() for a, b in zip(rl, g_loc))
     + sum((a * b.double()).sum() for a, b in zip(rn, g_only))).backward()
    # ---- device ----
    lg = [torch.zeros(B, x, x, 32).index_copy_(3, torch.arange(15), t.permute(0, 2, 3, 1)).to(dev).requires_grad_(True) for t, x in zip(logits, grids)]
    sg = [t.to(dev).requires_grad_(True) for t in sim]; qg = q_loc.to(dev).requires_grad_(True)
    pg = {k: v.to(dev).requires_grad_(True) for k, v in prm.items()}
    bn_le.to(dev); bn_lt.to(dev)
    res = HeadTail.apply(*lg, *sg, qg, coord.to(dev), pg["w_le"], pg["b_le"], pg["g_le"], pg["be_le"], pg["w_lt"], pg["b_lt"], pg["g_lt"],
                         pg["be_lt"], bn_le, bn_lt, training)
    po, pl, pn = res[0:3], res[3:6], res[6:9]
    (sum((a * b.to(dev)).sum() for a, b in zip(po, g_out)) + sum((a * b.to(dev)).sum() for a, b in zip(pl, g_loc))
     + sum((a * b.to(dev)).sum() for a, b in zip(pn, g_only))).backward()
    for s in range(3):
        _close(pn[s], rn[s], 2e-6, "only_obj"); _close(pl[s], rl[s], 2e-4, "loc_score"); _close(po[s], ro[s], 2e-4, "outbox")
    if training:
        _close(bn_le.running_mean, st["rm_le"], 1e-5, "rm_le"); _close(bn_le.running_var, st["rv_le"], 1e-5, "rv_le")
        _close(bn_lt.running_mean, st["rm_lt"], 1e-5, "rm_lt"); _close(bn_lt.running_var, st["rv_lt"], 1e-4, "rv_lt")
    # gradients: the min-max normalisation amplifies rounding, so compare directionally and at 2e-3 of the scale
    def chk(a, b, name):
        a = a.detach().cpu().double().flatten(); b = b.detach().double().flatten()
        if float(b.abs().max()) < 1e-9:
            assert float(a.abs().max()) < 1e-4, name
            return
        cos = float(F.cosine_similarity(a, b, dim=0))
        rel = float((a - b).abs().max() / b.abs().max())
        assert cos > 0.9999 and rel < 5e-3, (name, cos, rel)
    for s in range(3):
        chk(lg[s].grad[..., :15].permute(0, 3, 1, 2), ld[s].grad, f"dlogits{s}"); chk(sg[s].grad, sd[s].grad, f"dsim{s}")
        assert float(lg[s].grad[..., 15:].abs().max()) == 0.0
    chk(qg.grad, qd.grad, "dq_loc")
    for k in prm:
        if training and k in ("b_le", "b_lt"):
            assert float(pg[k].grad.abs().max()) < 1e-3 * max(1.0, float(pd["w_lt"].grad.abs().max())), k   # a bias before batch-stat BN
            continue
        chk(pg[k].grad, pd[k].grad, k)


# ---- K9 -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,g_,e", [(4, 8, 512), (6, 13, 512), (2, 19, 512)])
def test_interframe_sampling_matches_torch(dev, n, g_, e):
    from dcnet_amd import ops
    from dcnet_amd.functions import InterframeSample
    g = _g(n + g_)
    hw, top_k, neg_n = g_ * g_, 30, 10
    fv = F.normalize(torch.randn(n, g_, g_, e, generator=g), dim=3)
    raw = torch.stack([torch.stack([torch.randperm(hw - 1, generator=g)[:neg_n] for _ in range(top_k)]) for _ in range(n // 2)])
    fg = fv.to(dev).requires_grad_(True)
    frame, corr, negf, index, neg_idx = InterframeSample.apply(fg, raw.to(dev), top_k)
    f = fv.view(n // 2, 2, hw, e)
    # the selection: exactly torch.topk on the affinity the device computed (:390-395)
    cmap = ops.gemm_nt_batched(fg.detach().view(n // 2, 2, hw, e)[:, 0], fg.detach().view(n // 2, 2, hw, e)[:, 1]).flatten(1)
    _close(cmap, torch.bmm(f[:, 0].double(), f[:, 1].double().transpose(1, 2)).flatten(1), 3e-6, "cmap")
    tv, ti = cmap.topk(top_k, dim=1, largest=True, sorted=True)
    assert torch.equal(torch.gather(cmap, 1, index), tv)                       # same values in the same (descending) order
    assert torch.equal(index, ti) or bool((tv[:, 1:] == tv[:, :-1]).any())     # and the same indices unless values tie
    qi, ki = index.cpu() // hw, index.cpu() % hw                               # :407,409
    ni = raw + (raw >= ki.unsqueeze(2)).long()                                 # :411-413
    assert torch.equal(neg_idx.cpu(), ni)
    ar = torch.arange(n // 2)
    fd = fv.double().view(n // 2, 2, hw, e).requires_grad_(True)
    r_frame = fd[:, 0][ar.unsqueeze(1), qi]; r_corr = fd[:, 1][ar.unsqueeze(1), ki]; r_neg = fd[:, 1][ar.view(-1, 1, 1), ni]
    assert torch.equal(frame.cpu(), r_frame.float()) and torch.equal(corr.cpu(), r_corr.float()) and torch.equal(negf.cpu(), r_neg.float())
    g1, g2, g3 = torch.randn(frame.shape, generator=g), torch.randn(corr.shape, generator=g), torch.randn(negf.shape, generator=g)
    ((r_frame * g1.double()).sum() + (r_corr * g2.double()).sum() + (r_neg * g3.double()).sum()).backward()
    ((frame * g1.to(dev)).sum() + (corr * g2.to(dev)).sum() + (negf * g3.to(dev)).sum()).backward()
    _close(fg.grad.view(n // 2, 2, hw, e), fd.grad, 2e-6, "dfv")


def test_topk_ties_take_the_lowest_index_first(dev):
    """Degenerate affinities (many equal values): the selection is still a valid sorted top-k, ties resolved by flat index."""
    from dcnet_amd.lib import lib
    hw, e, top_k, neg_n = 9, 32, 30, 3
    cmap = torch.zeros(2, hw * hw)
    cmap[0, 5] = 2.0; cmap[0, 17] = 2.0; cmap[0, 3] = 3.0; cmap[0, 40:70] = 1.0       # 3 above, 30 tied at 1.0: take 40..66
    cmap[1] = -1.0                                                                  # all equal: indices 0..29
    fv = torch.randn(4, hw, e); raw = torch.zeros(2, top_k, neg_n, dtype=torch.int64)
    index = torch.empty(2, top_k, dtype=torch.int64, device=dev); neg_idx = torch.empty(2, top_k, neg_n, dtype=torch.int64, device=dev)
    fr = torch.empty(2, top_k, e, device=dev); co = torch.empty_like(fr); ng = torch.empty(2, top_k, neg_n, e, device=dev)
    cm, fvd, rw = cmap.to(dev), fv.to(dev), raw.to(dev)
    lib().k9_fwd(cm.data_ptr(), fvd.data_ptr(), rw.data_ptr(), 2, hw, e, top_k, neg_n, index.data_ptr(), neg_idx.data_ptr(), fr.data_ptr(),
                 co.data_ptr(), ng.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert index[0].tolist() == [3, 5, 17] + list(range(40, 67))
    assert index[1].tolist() == list(range(30))


# ---- K14 ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,g_,L", [(4, 8, 20), (3, 13, 20)])
def test_crossmodal_sampling_matches_torch(dev, n, g_, L):
    import numpy as np
    from dcnet_amd import ops
    from dcnet_amd.functions import CrossModalSample
    from dcnet_amd.lib import lib
    g = _g(n * 3 + g_)
    hw, e, neg_n = g_ * g_, 512, 5
    fv = F.normalize(torch.randn(n, g_, g_, e, generator=g), dim=3)
    ctx = torch.randn(n, L, 2 * e, generator=g)
    cw = torch.randn(L, L, 3, generator=g) * 0.2; cb = torch.randn(L, generator=g) * 0.1
    neg = torch.randint(0, hw, (n, hw, neg_n), generator=g)
    off = np.zeros(hw + 1, dtype=np.int32); src = np.zeros(n * hw * neg_n, dtype=np.int32)
    lib().mt_sample_crossmodal_csr(neg.numpy().ctypes.data, n, hw, neg_n, off.ctypes.data, src.ctypes.data)
    # host-side inverse: every source exactly once, grouped by destination, ascending inside a group
    flat = neg.flatten().numpy()
    assert off[0] == 0 and off[-1] == flat.size and sorted(src.tolist()) == list(range(flat.size))
    for p in (0, hw // 2, hw - 1):
        seg = src[off[p]:off[p + 1]]
        assert (flat[seg] == p).all() and (np.diff(seg) > 0).all()
    fg, cg = fv.to(dev).requires_grad_(True), ctx.to(dev).requires_grad_(True)
    vit, lag_pos, neg_cross, cols = CrossModalSample.apply(fg, cg, cw.to(dev), cb.to(dev), neg.to(dev), torch.from_numpy(off).to(dev),
                                                           torch.from_numpy(src).to(dev))
    fd, cd = fv.double().view(n, hw, e).requires_grad_(True), ctx.double().requires_grad_(True)
    r_vit = F.normalize(fd.permute(0, 2, 1), dim=2).permute(0, 2, 1)           # over positions (:629), back to (N,HW,E)
    r_lag = F.normalize(cd[:, :, 0::2], dim=1)                                 # interpolate(0.5) + over words (:631-632)
    lv = F.conv1d(torch.bmm(r_lag, r_vit.transpose(1, 2)), cw.double(), cb.double(), padding=1)   # :634-635
    _close(vit, r_vit, 2e-6, "vit")
    lv_dev = ops.crossmap(ops.lagnorm_fwd(cg.detach())[0], vit.detach(), cw.to(dev), cb.to(dev), want_map=True)[1]
    _close(lv_dev, lv, 5e-6, "conv map")
    picked = torch.gather(lv, 1, cols.cpu().unsqueeze(1)).squeeze(1)
    assert float((lv.max(dim=1)[0] - picked).abs().max()) < 1e-5               # the device's word IS an arg-max of the fp64 map
    ar = torch.arange(n)
    r_lp = r_lag[ar.unsqueeze(1), cols.cpu()].unsqueeze(2); r_nc = r_vit[n - 1][neg]     # :66-70, :75-96 (image N-1)
    _close(lag_pos, r_lp, 2e-6, "lag_pos"); _close(neg_cross, r_nc, 2e-6, "neg_cross")
    g1, g2, g3 = torch.randn(vit.shape, generator=g), torch.randn(lag_pos.shape, generator=g), torch.randn(neg_cross.shape, generator=g)
    ((r_vit * g1.double()).sum() + (r_lp * g2.double()).sum() + (r_nc * g3.double()).sum()).backward()
    ((vit * g1.to(dev)).sum() + (lag_pos * g2.to(dev)).sum() + (neg_cross * g3.to(dev)).sum()).backward()
    _close(fg.grad.view(n, hw, e), fd.grad, 2e-5, "dfv"); _close(cg.grad, cd.grad, 2e-5, "dcontext")


# ---- InfoNCE, row scores --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,m", [(60, 10), (676, 5), (7, 1)])
def test_contrastive_rows_fwd_bwd(dev, rows, m):
    from dcnet_amd.functions import Contrastive
    g = _g(rows + m)
    e, T = 512, 0.07
    q, k, neg = torch.randn(rows, e, generator=g), torch.randn(rows, e, generator=g), torch.randn(rows, m, e, generator=g)
    qd, kd, nd = (t.double().requires_grad_(True) for t in (q, k, neg))
    qn = F.normalize(qd, dim=1); kn = F.normalize(kd, dim=1); nn_ = F.normalize(nd.permute(0, 2, 1), dim=1)      # train_DCNet.py:120-124
    logits = torch.cat([torch.einsum("nc,nc->n", qn, kn).unsqueeze(-1), torch.einsum("nc,nck->nk", qn, nn_)], 1) / T
    ref = F.cross_entropy(logits, torch.zeros(rows, dtype=torch.long))
    (ref * 3.0).backward()
    qg, kg, ng = (t.to(dev).requires_grad_(True) for t in (q, k, neg))
    out = Contrastive.apply(qg, kg, ng, T)
    (out * 3.0).backward()
    assert abs(float(out) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    _close(qg.grad, qd.grad, 2e-5, "dq"); _close(kg.grad, kd.grad, 2e-5, "dk"); _close(ng.grad, nd.grad, 2e-5, "dneg")


def test_norm_score_flip_and_rowdot(dev):
    from dcnet_amd.functions import NormAccumulate, NormScore, RowDot
    g = _g(5)
    n, h, w, e = 4, 5, 6, 512
    x = torch.randn(n, h, w, e, generator=g); q = F.normalize(torch.randn(n, e, generator=g), dim=1)
    gc, gs, gn = torch.randn(n, h, w, e, generator=g), torch.randn(n, h, w, generator=g), torch.randn(n, h, w, generator=g)
    xd, qd = x.double().requires_grad_(True), q.double().requires_grad_(True)
    cr = F.normalize(xd, dim=3); sr = (cr * qd.view(n, 1, 1, e)).sum(3); nr = (cr * qd.flip(0).view(n, 1, 1, e)).sum(3)
    ((cr * gc.double()).sum() + (sr * gs.double()).sum() + (nr * gn.double()).sum()).backward()
    xg, qg = x.to(dev).requires_grad_(True), q.to(dev).requires_grad_(True)
    corr, sim, neg = NormScore.apply(xg, qg, True)
    ((corr * gc.to(dev)).sum() + (sim * gs.to(dev)).sum() + (neg * gn.to(dev)).sum()).backward()
    _close(corr, cr, 2e-6); _close(sim, sr, 2e-6); _close(neg, nr, 2e-6)
    _close(xg.grad, xd.grad, 1e-5, "dx"); _close(qg.grad, qd.grad, 1e-5, "dq")
    for flip in (False, True):
        xd2, qd2 = x.double().requires_grad_(True), q.double().requires_grad_(True)
        ref = (xd2 * (qd2.flip(0) if flip else qd2).view(n, 1, 1, e)).sum(3)
        (ref * gs.double()).sum().backward()
        xg2, qg2 = x.to(dev).requires_grad_(True), q.to(dev).requires_grad_(True)
        got = RowDot.apply(xg2, qg2, flip)
        (got * gs.to(dev)).sum().backward()
        _close(got, ref, 3e-6); _close(xg2.grad, xd2.grad, 1e-5); _close(qg2.grad, qd2.grad, 1e-5)
    with torch.no_grad():
        acc = NormAccumulate.apply(x.to(dev), None, 0.25)
        acc = NormAccumulate.apply((x * 2 + 1).to(dev), acc, 0.25)
    _close(acc, 0.25 * F.normalize(x.double(), dim=3) + 0.25 * F.normalize((x * 2 + 1).double(), dim=3), 2e-6)
