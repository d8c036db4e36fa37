"""autograd nodes of the DCNet head.  Each forward/backward is a sequence of libdcnet_hip.so
kernels (dcnet_amd.ops); torch only carries the tensors between them.  All feature maps are NHWC.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class ConvBNAct(torch.autograd.Function):
    """ConvBatchNormReLU (model/darknet.py:118-156): conv (no bias) + BatchNorm2d(momentum 0.999)
    + ReLU.  x may carry zero-padded channels beyond the weight's Cin (K-padding to 32)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, ksize: int, training: bool, slope: float, amax_x=None, bank=None, out_b16: bool = False):
        """Returns (out, amax_out): amax_* are the abs-max words of ops.amax_* (None outside the f16-split precision).
        bank: this layer's entry of an ops.FilterBanks table refreshed this step (OHWI / split / transposed banks + abs-max word):
        nothing is transposed, measured or split per launch then.
        bf16-storage mode (ops.storage_b16(), a bank given): the block runs on bf16 tensors — x bf16 (an fp32 x is cast once), raw result
        and saved tensors bf16, out bf16 when ``out_b16`` (the consumer is another bf16 block) else fp32, written by the BatchNorm pass
        itself; the input gradient comes back in x's dtype straight from the data-gradient kernel."""
        cout = weight.shape[0]
        am = ops.use_amax()
        if bank is not None and x.shape[3] != weight.shape[1]:
            bank = None                    # (K-padded input: the per-launch path pads the bank)
        ctx.b16 = None
        ctx.wparam = weight if (ops.WGRAD_DIRECT and ops.WGRAD_SIDE and training and isinstance(weight, torch.nn.Parameter)
                                and weight.requires_grad) else None
        if ops.storage_b16() and bank is not None:
            x_f32 = x.dtype == torch.float32
            x16 = ops.to_b16(x.contiguous())
            use8 = ops.f8_takes(x16.shape[3], cout, ksize)          # "fp8s": the head's 3x3 block on e4m3 operands
            if use8:
                x8, xs = ops.quant_of(x16)
                w8, ws = ops.bank_q8(bank, "q8", bank["b16"], cout)
            if training and use8:
                y, stats = ops.conv2d_fwd_f8(x8, xs, w8.view(-1), ws, cout, ksize, 1, want_stats=True)
                mi = ops.bn_finalize(stats, y.numel() // cout, gamma.detach(), beta.detach(), bn.eps, bn.momentum,
                                     bn.running_mean, bn.running_var)
                ops.bump_batches(bn)
                out = ops.scale_act(y, mi[2], mi[3], ops.ACT_LEAKY, slope, out_f32=not out_b16)
                ctx.save_for_backward(x16, y, mi, gamma, beta)
            elif training:
                y, stats = ops.conv2d_fwd_b16(x16, bank["b16"], cout, ksize, 1, want_stats=True)
                mi = ops.bn_finalize(stats, y.numel() // cout, gamma.detach(), beta.detach(), bn.eps, bn.momentum,
                                     bn.running_mean, bn.running_var)
                ops.bump_batches(bn)
                out = ops.scale_act(y, mi[2], mi[3], ops.ACT_LEAKY, slope, out_f32=not out_b16)
                ctx.save_for_backward(x16, y, mi, gamma, beta)
            else:
                ss = ops.bn_fold(gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, bn.eps)
                if use8:
                    out, _ = ops.conv2d_fwd_f8(x8, xs, w8.view(-1), ws, cout, ksize, 1, ss[0], ss[1], ops.ACT_LEAKY, slope, out_f32=not out_b16)
                else:
                    out, _ = ops.conv2d_fwd_b16(x16, bank["b16"], cout, ksize, 1, ss[0], ss[1], ops.ACT_LEAKY, slope, out_f32=not out_b16)
            ctx.b16 = (bank["tb16"], x_f32); ctx.bank = bank
            ctx.meta = (ksize, training, slope, tuple(weight.shape))
            return out, None
        wsp = wtr = None
        if bank is not None:
            w, wsp, wtr = bank["ohwi"], bank["split"], (bank["t"], bank["tsplit"])
            aw = bank["amax"] if am else None
        else:
            w = ops.weight_to_ohwi(weight, ci_pad=x.shape[3])
            aw = ops.absmax(weight.detach()) if am else None
        ax = (amax_x if amax_x is not None else ops.absmax(x)) if am else None
        ao = ops.amax_slot(x.device) if am else None
        ctx.wtr = wtr
        if training:
            y, stats = ops.conv2d_fwd(x, w, ksize, 1, want_stats=True, amax_x=ax, amax_w=aw, w_split_ready=wsp)
            mi = ops.bn_finalize(stats, y.numel() // cout, gamma.detach(), beta.detach(), bn.eps, bn.momentum,
                                 bn.running_mean, bn.running_var)
            ops.bump_batches(bn)
            out = ops.scale_act(y, mi[2], mi[3], ops.ACT_LEAKY, slope, amax_out=ao)
            ctx.save_for_backward(x, y, mi, w, gamma, beta)
        else:
            ss = ops.bn_fold(gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, bn.eps)
            out, _ = ops.conv2d_fwd(x, w, ksize, 1, ss[0], ss[1], ops.ACT_LEAKY, slope, amax_x=ax, amax_w=aw, amax_out=ao, w_split_ready=wsp)
            ctx.save_for_backward(x, out, ss, w, gamma, beta)
        ctx.meta = (ksize, training, slope, tuple(weight.shape))
        ctx.amax = (ax, aw)
        if ao is not None:
            ctx.mark_non_differentiable(ao)
        return out, ao

    @staticmethod
    def backward(ctx, dout, _ga=None):
        if ctx.b16 is not None:            # bf16 storage: dout bf16 or fp32 (read as it is), dy / saved tensors bf16, dw fp32
            if not ctx.meta[1]:
                raise NotImplementedError("ConvBNAct, bf16 storage: no backward in eval mode (frozen-BatchNorm fine-tuning is not built)")
            x16, y, mi, gamma, beta = ctx.saved_tensors
            ksize, training, slope, wshape = ctx.meta
            tb16, x_f32 = ctx.b16
            dy, dgamma, dbeta = ops.bn_act_bwd(y, dout.contiguous(), mi[0], mi[1], gamma.detach(), beta.detach(), ops.ACT_LEAKY, slope,
                                               quant=ctx.needs_input_grad[0] and ops.f8_takes(y.shape[3], x16.shape[3], ksize))
            dx = None

            def dgrad():
                if ops.f8_takes(dy.shape[3], x16.shape[3], ksize) and dy.is_contiguous():      # "fp8s": the data gradient on e4m3 operands
                    dy8, dys = ops.quant_of(dy)
                    wt8, wts = ops.bank_q8(getattr(ctx, "bank", None), "tq8", tb16, x16.shape[3])
                    return ops.conv2d_bwd_data_f8(dy8, dys, wt8.view(-1), wts, (x16.shape[1], x16.shape[2]), x16.shape[3], ksize, 1, out_f32=x_f32)
                return ops.conv2d_bwd_data_b16(dy, tb16, (x16.shape[1], x16.shape[2]), x16.shape[3], ksize, 1, out_f32=x_f32)

            if ctx.wparam is not None:     # (ops.WGRAD_DIRECT: the block before's weight gradient goes out behind this block's passes)
                ops.release_held_wgrads()
                if ctx.needs_input_grad[0]:
                    dx = dgrad()
                ops.hold_wgrad_into(ctx.wparam, x16, dy, ksize, 1, wshape)
                return dx, None, dgamma, dbeta, None, None, None, None, None, None, None
            if not ops.WGRAD_AFTER_DGRAD:
                dwt = ops.wgrad_on_side(x16, dy, ksize, 1, wshape)
            if ctx.needs_input_grad[0]:
                dx = dgrad()
            if ops.WGRAD_AFTER_DGRAD:
                dwt = ops.wgrad_on_side(x16, dy, ksize, 1, wshape)
            ops.join_side(x16.device)
            return dx, dwt, dgamma, dbeta, None, None, None, None, None, None, None
        x, y, aux, w, gamma, beta = ctx.saved_tensors
        gamma, beta = gamma.detach(), beta.detach()
        ksize, training, slope, wshape = ctx.meta
        ax, aw = ctx.amax
        dout = dout.contiguous()
        ady = None
        if training:
            ady = ops.amax_slot(dout.device) if ops.use_amax() else None
            dy, dgamma, dbeta = ops.bn_act_bwd(y, dout, aux[0], aux[1], gamma, beta, ops.ACT_LEAKY, slope, amax_out=ady)
        else:
            dz = ops.act_bwd(y, dout, slope)
            dy = dz * aux[0]
            dbeta = dz.reshape(-1, wshape[0]).sum(0)
            z = y if slope == 0 else torch.where(y > 0, y, y / slope)
            gs = torch.where(gamma == 0, torch.ones_like(gamma), gamma)
            dgamma = (dz * (z - beta) / gs).reshape(-1, wshape[0]).sum(0)
        if ctx.wparam is not None and training:
            ops.release_held_wgrads()
            dx = ops.conv2d_bwd_data(dy, w, (x.shape[1], x.shape[2]), ksize, 1, amax_dy=ady, amax_w=aw, wt_ready=ctx.wtr) if ctx.needs_input_grad[0] else None
            ops.hold_wgrad_into(ctx.wparam, x, dy, ksize, 1, wshape, amax_x=ax, amax_dy=ady)
            return dx, None, dgamma, dbeta, None, None, None, None, None, None, None
        if not ops.WGRAD_AFTER_DGRAD:
            dwt = ops.wgrad_on_side(x, dy, ksize, 1, wshape, amax_x=ax, amax_dy=ady)
        dx = ops.conv2d_bwd_data(dy, w, (x.shape[1], x.shape[2]), ksize, 1, amax_dy=ady, amax_w=aw, wt_ready=ctx.wtr) if ctx.needs_input_grad[0] else None
        if ops.WGRAD_AFTER_DGRAD:
            dwt = ops.wgrad_on_side(x, dy, ksize, 1, wshape, amax_x=ax, amax_dy=ady)
        ops.join_side(x.device)
        return dx, dwt, dgamma, dbeta, None, None, None, None, None, None, None


class ConvBias(torch.autograd.Function):
    """Plain 1x1 conv with bias (the 256 -> 15 bbox head, model/DCNet_model.py:331).  The filter
    bank is zero-padded to 32 outputs so the data gradient's contraction is MFMA-aligned; the
    caller slices [..., :cout]."""

    @staticmethod
    def forward(ctx, x, weight, bias, amax_x=None):
        cout = weight.shape[0]
        cop = ops.pad32(cout)
        ctx.wparam = weight if (ops.WGRAD_DIRECT and ops.WGRAD_SIDE and isinstance(weight, torch.nn.Parameter) and weight.requires_grad) else None
        w = ops.weight_to_ohwi(weight, ci_pad=x.shape[3], co_pad=cop)
        b = torch.zeros(cop, dtype=torch.float32, device=x.device)
        b[:cout] = bias.detach()
        ctx.b16 = None
        if ops.storage_b16() and weight.shape[2] == 1 and x.shape[3] % 32 == 0:
            # bf16 storage: x bf16 (an fp32 x is cast once), the 32 x Cin bank in bf16 (8 K elements: converted here), logits fp32
            x_f32 = x.dtype == torch.float32
            x16 = ops.to_b16(x.contiguous())
            w16 = w.reshape(cop, -1).to(torch.bfloat16)
            y, _ = ops.conv2d_fwd_b16(x16, w16.reshape(-1), cop, 1, 1, None, b, out_f32=True)
            ctx.save_for_backward(x16, w16)
            ctx.wshape = tuple(weight.shape)
            ctx.b16 = x_f32
            return y
        y, _ = ops.conv2d_fwd(x, w, weight.shape[2], 1, None, b)          # (32 filters: the fp32-pipe tile, no scales)
        ctx.save_for_backward(x, w)
        ctx.wshape = tuple(weight.shape)
        ctx.amax = (amax_x, ops.absmax(weight.detach()) if ops.use_amax() else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.b16 is not None:
            x16, w16 = ctx.saved_tensors
            co, ci, k, _ = ctx.wshape
            dy = dy.contiguous()
            db = ops.colsum_rows(dy.view(-1, dy.shape[-1]))[:co]
            dy16 = ops.to_b16(dy)
            dx = None
            if ctx.wparam is not None:      # (ops.WGRAD_DIRECT: added to .grad on the side stream, no join here)
                ops.release_held_wgrads()
                if ctx.needs_input_grad[0]:
                    dx = ops.conv2d_bwd_data_b16(dy16, w16.t().contiguous().reshape(-1), (x16.shape[1], x16.shape[2]), x16.shape[3], 1, 1,
                                                 out_f32=ctx.b16)
                ops.hold_wgrad_into(ctx.wparam, x16, dy16, 1, 1, ctx.wshape)
                return dx, None, db, None
            dwt = ops.wgrad_on_side(x16, dy16, 1, 1, ctx.wshape)
            if ctx.needs_input_grad[0]:
                dx = ops.conv2d_bwd_data_b16(dy16, w16.t().contiguous().reshape(-1), (x16.shape[1], x16.shape[2]), x16.shape[3], 1, 1,
                                             out_f32=ctx.b16)
            ops.join_side(x16.device)
            return dx, dwt, db, None
        x, w = ctx.saved_tensors
        co, ci, k, _ = ctx.wshape
        ax, aw = ctx.amax
        dy = dy.contiguous()
        ady = ops.absmax(dy) if ops.use_amax() else None
        if ctx.wparam is not None:
            ops.release_held_wgrads()
            db = ops.colsum_rows(dy.view(-1, dy.shape[-1]))[:co]
            dx = ops.conv2d_bwd_data(dy, w, (x.shape[1], x.shape[2]), k, 1, amax_dy=ady, amax_w=aw) if ctx.needs_input_grad[0] else None
            ops.hold_wgrad_into(ctx.wparam, x, dy, k, 1, ctx.wshape, amax_x=ax, amax_dy=ady)
            return dx, None, db, None
        dwt = ops.wgrad_on_side(x, dy, k, 1, ctx.wshape, amax_x=ax, amax_dy=ady)
        db = ops.colsum_rows(dy.view(-1, dy.shape[-1]))[:co]
        dx = ops.conv2d_bwd_data(dy, w, (x.shape[1], x.shape[2]), k, 1, amax_dy=ady, amax_w=aw) if ctx.needs_input_grad[0] else None
        ops.join_side(x.device)
        return dx, dwt, db, None


class L2Norm(torch.autograd.Function):
    """F.normalize(x, p=2, dim=channel) on an NHWC map (model/DCNet_model.py:359)."""

    @staticmethod
    def forward(ctx, x):
        out, norm, _, _ = ops.l2norm_score_fwd(x)
        ctx.save_for_backward(out, norm)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, norm = ctx.saved_tensors
        dx, _ = ops.l2norm_score_bwd(out, norm, dout.contiguous(), None, None, 0)
        return dx


class NormScore(torch.autograd.Function):
    """One pass over the corr_conv output x (N,H,W,E):  corr = x / ||x||  (model/DCNet_model.py:469), sim = <corr, q>
    (:530-535, q = flang_attn) and, with ``want_neg``, neg_sim = <corr, q reversed along the batch> — the caller's
    neg_sim_score (train_DCNet.py:623-627) from the same read.  Returns (corr (N,H,W,E), sim (N,H,W), neg_sim|None)."""

    @staticmethod
    def forward(ctx, x, q, want_neg: bool):
        n, h, w, e = x.shape
        q = q.contiguous()
        corr, norm, score, flip = ops.l2norm_score_fwd(x, q, h * w, want_flip=want_neg)
        ctx.save_for_backward(corr, norm, q)
        if not want_neg:
            return corr, score.view(n, h, w), None
        return corr, score.view(n, h, w), flip.view(n, h, w)

    @staticmethod
    def backward(ctx, dcorr, dscore, dneg):
        corr, norm, q = ctx.saved_tensors
        q = q.detach()
        n, h, w, _ = corr.shape
        dout = dcorr.contiguous() if dcorr is not None else None
        ds = dscore.contiguous().view(-1) if dscore is not None else None
        dn = dneg.contiguous().view(-1) if dneg is not None else None
        use_q = q if (ds is not None or dn is not None) else None
        dx, dq = ops.l2norm_score_bwd(corr, norm, dout, use_q, ds, h * w, dscore_flip=dn)
        return dx, dq, None


class RowDot(torch.autograd.Function):
    """score (N,H,W) = <x[n,h,w,:], q[n]> (flip: q[N-1-n]) on a map that is NOT normalised here: sim_score of the n_frame
    model (model/test_DCNet_model.py:386-391) and neg_sim_score on plain tensors (train_DCNet.py:623-627)."""

    @staticmethod
    def forward(ctx, x, q, flip: bool):
        n, h, w, e = x.shape
        x = x.contiguous(); q = q.contiguous()
        ctx.save_for_backward(x, q)
        ctx.flip = flip
        return ops.rowdot_fwd(x, q, h * w, flip).view(n, h, w)

    @staticmethod
    def backward(ctx, ds):
        x, q = ctx.saved_tensors
        n, h, w, e = x.shape
        dx, dq = ops.rowdot_bwd(x, q.detach(), ds.contiguous().view(-1), h * w, ctx.flip, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dq, None


class NormAccumulate(torch.autograd.Function):
    """acc (+)= scale * normalize(x) — the mean over the T-1 normalised correspondence features of the inference model
    (model/test_DCNet_model.py:277-280,324-332).  Forward only, like the rest of the n_frame path."""

    @staticmethod
    def forward(ctx, x, acc: Optional[torch.Tensor], scale: float):
        out, _, _, _ = ops.l2norm_score_fwd(x, out=acc, out_scale=scale, accumulate=acc is not None)
        ctx.mark_dirty(*([acc] if acc is not None else []))
        return out

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("the n_frame (inference) model has no backward; train with pair semantics")


class PhraseAttn(torch.autograd.Function):
    """PhraseAttention (model/DCNet_model.py:190-219) for one or two heads over the same (context, embedded, ids), each
    followed by F.normalize when ``normalize`` (:526,:557).  Returns (out_0, out_1|None, attn (H,N,L))."""

    @staticmethod
    def forward(ctx, context, embedded, ids, w0, b0, w1, b1, normalize: bool):
        context = context.contiguous(); embedded = embedded.contiguous(); ids = ids.contiguous()
        d = context.shape[2]
        wa = w0.detach().reshape(-1).contiguous(); wb = None if w1 is None else w1.detach().reshape(-1).contiguous()
        attn, out, vnorm = ops.phrase_attn_fwd(context, embedded, ids, wa, b0.detach(), wb, None if b1 is None else b1.detach(), normalize)
        ctx.save_for_backward(context, embedded, wa, wb if wb is not None else wa.new_empty(0), attn, out,
                              vnorm if vnorm is not None else wa.new_empty(0))
        ctx.meta = (normalize, w1 is not None, tuple(w0.shape), d)
        ctx.mark_non_differentiable(attn)
        return out[0], (out[1] if w1 is not None else None), attn

    @staticmethod
    def backward(ctx, g0, g1, _gattn):
        context, embedded, wa, wb, attn, out, vnorm = ctx.saved_tensors
        normalize, two, wshape, d = ctx.meta
        dout = torch.zeros_like(out) if (g0 is None or (two and g1 is None)) else torch.empty_like(out)
        if g0 is not None:
            ops.copy_slice(g0.contiguous(), dout[0])
        if two and g1 is not None:
            ops.copy_slice(g1.contiguous(), dout[1])
        dctx, demb, dwb = ops.phrase_attn_bwd(context, embedded, wa, wb if two else None, attn, out, vnorm if normalize else None, dout, normalize)
        h = 2 if two else 1
        dw0 = dwb[:d].view(wshape); db0 = dwb[h * d:h * d + 1]
        dw1 = dwb[d:2 * d].view(wshape) if two else None
        db1 = dwb[2 * d + 1:2 * d + 2] if two else None
        return dctx, demb, None, dw0, db0, dw1, db1, None


class HeadTail(torch.autograd.Function):
    """The cross-scale tail (model/DCNet_model.py:545-621): only_obj, the location module in its rank-8 form
    (csrc/head.hip, csrc/locmod.hip), min-max normalisation, confidence modulation, NCHW outbox.

    Inputs: logits[3] (B,H,W,32) NHWC from fcn_out, sim[3] (B,H,W), q_loc (B,512) the normalised loc_attn phrase vector,
    coord (P,8) the concatenated coordinate rows, then the parameters of loc_embedding and loc_text_embedding.
    Returns outbox[3] (B,15,H,W), loc_score[3] (B,H,W), only_obj[3] (B,H,W)."""

    @staticmethod
    def forward(ctx, l0, l1, l2, s0, s1, s2, q_loc, coord, w_le, b_le, g_le, be_le, w_lt, b_lt, g_lt, be_lt, bn_le, bn_lt, training: bool):
        logits = [l0.contiguous(), l1.contiguous(), l2.contiguous()]
        B = logits[0].shape[0]
        sims = [s.contiguous().view(B, -1) for s in (s0, s1, s2)]
        P = coord.shape[0]
        cnt = B * P
        det = lambda t: t.detach().contiguous()
        w_le, b_le, g_le, be_le, w_lt, b_lt, g_lt, be_lt = map(det, (w_le, b_le, g_le, be_le, w_lt, b_lt, g_lt, be_lt))
        q_loc = q_loc.contiguous()
        e8, xh, stat, mom = ops.locemb_fwd(coord, w_le, b_le, g_le, be_le, bn_le, training, cnt)          # :572-578
        only, obj_map, objn, X = ops.head_obj(logits, sims, e8)                                           # :545-569
        wp = ops.pad_rows(w_lt, X.shape[1])                                                               # (512, Ppad)
        M = ops.gemm_nt(X, wp).view(B, 8, -1)                                                             # :581-585, rank-8
        Mp, bp, saved = ops.locbn_fwd(M, mom, b_lt, g_lt, be_lt, bn_lt, training, cnt)
        loc_map = ops.locmod_fwd(e8, Mp, bp, q_loc)                                                       # :585-594
        outbox, loc, mm = ops.head_final_fwd(logits, sims, loc_map)                                       # :597-621
        if training:
            ops.bump_batches(bn_le); ops.bump_batches(bn_lt)
        ctx.save_for_backward(*logits, *sims, *loc, *only, q_loc, coord, w_le, g_le, be_le, b_lt, g_lt, e8, xh, stat, mom, obj_map, objn, X,
                              wp, M, Mp, bp, saved, mm, bn_lt.running_mean)
        ctx.training = training
        ctx.wshape = tuple(w_lt.shape)
        return (*outbox, *loc, *only)

    @staticmethod
    def backward(ctx, do0, do1, do2, dl0, dl1, dl2, dn0, dn1, dn2):
        sv = ctx.saved_tensors
        logits, sims, loc, only = list(sv[0:3]), list(sv[3:6]), list(sv[6:9]), list(sv[9:12])
        (q_loc, coord, w_le, g_le, be_le, b_lt, g_lt, e8, xh, stat, mom, obj_map, objn, X, wp, M, Mp, bp, saved, mm, rmean_lt) = sv[12:]
        B, P = obj_map.shape
        cnt = B * P
        training = ctx.training
        cg = lambda t: None if t is None else t.contiguous()
        d_out = [cg(do0), cg(do1), cg(do2)]; d_loc = [cg(dl0), cg(dl1), cg(dl2)]; d_only = [cg(dn0), cg(dn1), cg(dn2)]
        dloc_map = ops.head_dloc(logits, sims, loc, d_out, d_loc, mm)
        dE_part, dsum = ops.locmod_bwd(e8, Mp, bp, q_loc, dloc_map)          # dE_part (B,P,8); dsum (B,10,512): dMp | dbp_n | dq
        dE_a = ops.colsum(dE_part.view(B, P * 8)).view(P, 8)
        dMp = dsum[:, :8]                                                    # strided views: no copies
        dbp = ops.colsum(dsum[:, 8])
        dq_loc = dsum[:, 9]
        dM, gout, dmom = ops.locbn_bwd(M, mom, b_lt, g_lt, rmean_lt, saved, dMp, dbp, training, cnt)
        dM2 = dM.view(B * 8, -1)
        dX = ops.gemm_nn(dM2, wp)                                            # (B*8, Ppad)
        dwp = ops.gemm_tn(dM2, X)                                            # (512, Ppad)
        dw_lt = ops.pad_rows(dwp, P)
        dobj, dE_b = ops.head_fold(dX, e8, obj_map)
        dlogits, dsim = ops.head_dlogits(logits, sims, loc, only, d_out, d_only, obj_map, objn, dobj)
        g88 = ops.locemb_bwd(coord, w_le, g_le, be_le, xh, stat, dE_a, dE_b, dmom, training)
        shp = [t.shape for t in logits]
        dsim = [d.view(s_[0], s_[1], s_[2]) for d, s_ in zip(dsim, shp)]
        return (*dlogits, *dsim, dq_loc, None, g88[:64].view(8, 8), g88[64:72], g88[72:80], g88[80:88],
                dw_lt.view(ctx.wshape), gout[2], gout[0], gout[1], None, None, None)


class InterframeSample(torch.autograd.Function):
    """K9 (model/DCNet_model.py:381-430): top-30 matches of each frame pair's affinity + gathers, on the scale-0 map.
    fv (N,H,W,E), raw_neg (N/2,top_k,neg_n) host-drawn list positions.  Returns (frame (b,K,E), corr (b,K,E),
    negf (b,K,neg_n,E), index (b,K), neg_idx (b,K,neg_n))."""

    @staticmethod
    def forward(ctx, fv, raw_neg, top_k: int):
        n, h, w, e = fv.shape
        f = fv.contiguous().view(n, h * w, e)
        index, neg_idx, frame, corr, negf = ops.k9_fwd(f, raw_neg, top_k)
        ctx.save_for_backward(index, neg_idx)
        ctx.shape = (n, h, w, e)
        ctx.mark_non_differentiable(index, neg_idx)
        return frame, corr, negf, index, neg_idx

    @staticmethod
    def backward(ctx, d_frame, d_corr, d_neg, _a, _b):
        index, neg_idx = ctx.saved_tensors
        n, h, w, e = ctx.shape
        z = lambda g, like: torch.zeros(like, dtype=torch.float32, device=index.device) if g is None else g.contiguous()
        b, k, m = neg_idx.shape
        dfv = ops.k9_bwd(index, neg_idx, z(d_frame, (b, k, e)), z(d_corr, (b, k, e)), z(d_neg, (b, k, m, e)), h * w)
        return dfv.view(n, h, w, e), None, None


class CrossModalSample(torch.autograd.Function):
    """K14 (model/DCNet_model.py:625-637 + Crossmodal_corrspondence :41-112).  fv (N,H,W,E), context (N,L,2E),
    conv_w (L,L,3), conv_b (L), neg (N,HW,neg_n) positions in image N-1, csr_off/csr_src their inverse.
    Returns (vit (N,HW,E), lag_pos (N,HW,1,E), neg_cross (N,HW,neg_n,E), cols (N,HW))."""

    @staticmethod
    def forward(ctx, fv, context, conv_w, conv_b, neg, csr_off, csr_src):
        n, h, w, e = fv.shape
        v = fv.contiguous().view(n, h * w, e)
        context = context.contiguous()
        vit, cn = ops.colnorm_fwd(v)                                          # :629
        lag, ln = ops.lagnorm_fwd(context)                                    # :631-632
        cols, _ = ops.crossmap(lag, vit, conv_w.detach().contiguous(), conv_b.detach().contiguous())   # :634-635, :48
        lag_pos, neg_cross = ops.k14_gather(lag, vit, cols, neg)
        ctx.save_for_backward(vit, cn, lag, ln, cols, csr_off, csr_src)
        ctx.shape = (n, h, w, e)
        ctx.mark_non_differentiable(cols)
        return vit, lag_pos, neg_cross, cols

    @staticmethod
    def backward(ctx, d_vit, d_lag_pos, d_neg, _c):
        vit, cn, lag, ln, cols, csr_off, csr_src = ctx.saved_tensors
        n, h, w, e = ctx.shape
        hw = h * w
        dq = torch.zeros_like(vit) if d_vit is None else d_vit.contiguous()
        extra = ops.k14_negscatter(d_neg.contiguous(), csr_off, csr_src, hw) if d_neg is not None else None
        dv = ops.colnorm_bwd(vit, cn, dq, extra, n - 1)
        dctx = None
        if d_lag_pos is not None and ctx.needs_input_grad[1]:
            dctx = ops.k14_dlag(lag, ln, cols, d_lag_pos.contiguous())
        return dv.view(n, h, w, e), dctx, None, None, None, None, None


class Contrastive(torch.autograd.Function):
    """mean over rows of CE([cos(q,k), cos(q,neg_1..m)] / T, 0): Interframe_contrastive_loss / Crossmodal_constrastive_loss
    (train_DCNet.py:114-166) on stacked rows."""

    @staticmethod
    def forward(ctx, q, pos, neg, temperature: float):
        q, pos, neg = q.contiguous(), pos.contiguous(), neg.contiguous()
        ctx.save_for_backward(q, pos, neg)
        ctx.t = temperature
        return ops.contrastive_fwd(q, pos, neg, temperature)

    @staticmethod
    def backward(ctx, g):
        q, pos, neg = ctx.saved_tensors
        dq, dpos, dneg = ops.contrastive_bwd(q, pos, neg, ctx.t, g.contiguous().view(1))
        return dq, dpos, dneg, None


class DenseLosses(torch.autograd.Function):
    """(yolo_loss, rank_loss, loc_loss) of train_DCNet.py:45-72,173-220 from the NCHW outbox, the sim / neg_sim / loc maps
    and the compact targets of ops.build_target."""

    @staticmethod
    def forward(ctx, ob0, ob1, ob2, s0, s1, s2, n0, n1, n2, l0, l1, l2, ti, tf, size: int):
        c = lambda t: t.contiguous()
        outbox = [c(ob0), c(ob1), c(ob2)]; sim = [c(s0), c(s1), c(s2)]; neg = [c(n0), c(n1), c(n2)]; loc = [c(l0), c(l1), c(l2)]
        out, lse = ops.dense_loss_fwd(outbox, sim, neg, loc, ti, tf, size)
        ctx.save_for_backward(*outbox, *sim, *neg, *loc, ti, tf, lse)
        ctx.size = size
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, gy, gr, gl):
        sv = ctx.saved_tensors
        outbox, sim, neg, loc = list(sv[0:3]), list(sv[3:6]), list(sv[6:9]), list(sv[9:12])
        ti, tf, lse = sv[12:]
        z = torch.zeros((), dtype=torch.float32, device=ti.device)
        gout = torch.stack([z if g is None else g.reshape(()) for g in (gy, gr, gl)])
        d_ob, d_sim, d_ns, d_loc = ops.dense_loss_bwd(outbox, sim, neg, loc, ti, tf, lse, gout, ctx.size)
        return (*d_ob, *d_sim, *d_ns, *d_loc, None, None, None)


class FusionConvBNAct(torch.autograd.Function):
    """First fcn_emb block (model/DCNet_model.py:491-505) with the constant channels folded away.  The
    reference convolves the concat [corr (E) | tile(flang) (E) | coord (8)] with a 1x1 filter bank W = [W1|W2|W3];
    the tiled language vector and the coordinate map are constant over positions / images, so
        conv = W1 . corr[n,p] + (W2 . flang[n]) + (W3 . coord[p])
    i.e. a K = E convolution accumulated onto a pre-filled per-image + per-position term — half the MFMA work
    of the 1032-channel form, and no (N,H,W,1032) tensor.  Then BatchNorm2d (momentum 0.999) + ReLU."""

    @staticmethod
    def forward(ctx, corr, flang, coord, weight, gamma, beta, bn, training: bool, amax_x=None):
        n, h, w, e = corr.shape
        am = ops.use_amax()
        ax = (amax_x if amax_x is not None else ops.absmax(corr)) if am else None
        ao = ops.amax_slot(corr.device) if am else None
        co = weight.shape[0]
        wd = weight.detach().view(co, -1)
        w1 = wd[:, :e].contiguous().view(co, e, 1, 1)
        w2, w3 = wd[:, e:2 * e], wd[:, 2 * e:]
        coord2d = coord.reshape(h * w, -1).contiguous()
        # the pre-filled term W2.flang[n] + W3.coord[p] (csrc/fusion.hip; W2, W3 are column slices of the parameter: no copies)
        y = ops.fusion_prefill(ops.gemm_nt(flang.detach().contiguous(), w2), coord2d, w3).view(n, h, w, co)
        ctx.b16 = False
        if ops.storage_b16() and training and e % 32 == 0 and co % 32 == 0:
            # bf16 storage: the K = E convolution on bf16 operands (corr cast once, the 512 x 512 bank converted here), accumulated onto the
            # fp32 pre-fill in the kernel's epilogue — raw result and statistics fp32 —, the activation written as bf16 for the next block
            corr16 = ops.to_b16(corr)
            w16 = w1.view(co, e).to(torch.bfloat16)
            y, stats = ops.conv2d_fwd_b16(corr16, w16.reshape(-1), co, 1, 1, out=y, want_stats=True, accumulate=True, out_f32=True)
            mi = ops.bn_finalize(stats, n * h * w, gamma.detach(), beta.detach(), bn.eps, bn.momentum, bn.running_mean, bn.running_var)
            ops.bump_batches(bn)
            out = ops.scale_act(y, mi[2], mi[3], ops.ACT_LEAKY, 0.0, out_b16=True)
            ctx.save_for_backward(corr16, y, mi, w16, gamma, beta, flang, weight)
            ctx.training, ctx.coord2d, ctx.b16, ctx.amax = True, coord2d, True, (None, None)
            return out, None
        wk = ops.weight_to_ohwi(w1)
        aw = ops.absmax(wk) if am else None
        if training:
            y, stats = ops.conv2d_fwd(corr, wk, 1, 1, out=y, want_stats=True, accumulate=True, amax_x=ax, amax_w=aw)
            mi = ops.bn_finalize(stats, n * h * w, gamma.detach(), beta.detach(), bn.eps, bn.momentum, bn.running_mean, bn.running_var)
            ops.bump_batches(bn)
            out = ops.scale_act(y, mi[2], mi[3], ops.ACT_LEAKY, 0.0, amax_out=ao)
            ctx.save_for_backward(corr, y, mi, wk, gamma, beta, flang, weight)
        else:
            ss = ops.bn_fold(gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, bn.eps)
            out, _ = ops.conv2d_fwd(corr, wk, 1, 1, ss[0], ss[1], ops.ACT_LEAKY, 0.0, out=y, accumulate=True, amax_x=ax, amax_w=aw, amax_out=ao)
            ctx.save_for_backward(corr, out, ss, wk, gamma, beta, flang, weight)
        ctx.training = training
        ctx.coord2d = coord2d
        ctx.amax = (ax, aw)
        if ao is not None:
            ctx.mark_non_differentiable(ao)
        return out, ao

    @staticmethod
    def backward(ctx, dout, _ga=None):
        corr, y, aux, wk, gamma, beta, flang, weight = ctx.saved_tensors
        gamma, beta = gamma.detach(), beta.detach()
        n, h, w, e = corr.shape
        co = weight.shape[0]
        ax, aw = ctx.amax
        dout = dout.contiguous()
        if ctx.b16:
            dy16, dgamma, dbeta = ops.bn_act_bwd(y, dout, aux[0], aux[1], gamma, beta, ops.ACT_LEAKY, 0.0)      # y fp32, dout bf16 | fp32 -> dy bf16
            dw1 = ops.wgrad_on_side(corr, dy16, 1, 1, (co, e, 1, 1)).view(co, e)
            dcorr = ops.conv2d_bwd_data_b16(dy16, wk.t().contiguous().reshape(-1), (h, w), e, 1, 1, out_f32=True) if ctx.needs_input_grad[0] else None
            ops.join_side(corr.device)
            wd = weight.detach().view(co, -1)
            dweight = torch.empty((co, wd.shape[1]), dtype=torch.float32, device=dy16.device)
            d_img = ops.fusion_bwd(ops.to_f32(dy16), ctx.coord2d, flang.detach().contiguous(), dweight, e)     # (the per-image / per-position terms read fp32)
            ops.copy_slice(dw1, dweight[:, :e])
            dflang = ops.gemm_nn(d_img, wd[:, e:2 * e]) if ctx.needs_input_grad[1] else None
            return dcorr, dflang, None, dweight.view_as(weight), dgamma, dbeta, None, None, None
        ady = None
        if ctx.training:
            ady = ops.amax_slot(dout.device) if ops.use_amax() else None
            dy, dgamma, dbeta = ops.bn_act_bwd(y, dout, aux[0], aux[1], gamma, beta, ops.ACT_LEAKY, 0.0, amax_out=ady)
        else:
            dz = ops.act_bwd(y, dout, 0.0)
            dy = dz * aux[0]
            dbeta = dz.reshape(-1, co).sum(0)
            gs = torch.where(gamma == 0, torch.ones_like(gamma), gamma)
            dgamma = (dz * (y - beta) / gs).reshape(-1, co).sum(0)
        wd = weight.detach().view(co, -1)
        dw1 = ops.wgrad_on_side(corr, dy, 1, 1, (co, e, 1, 1), amax_x=ax, amax_dy=ady).view(co, e)      # overlaps with the data gradient
        dcorr = ops.conv2d_bwd_data(dy, wk, (h, w), 1, 1, amax_dy=ady, amax_w=aw) if ctx.needs_input_grad[0] else None
        ops.join_side(corr.device)
        # gradients of the per-image / per-position terms: one pass over dy (d_img, dW3), dW2 = d_img^T.flang, dflang = d_img.W2
        dweight = torch.empty((co, wd.shape[1]), dtype=torch.float32, device=dy.device)
        d_img = ops.fusion_bwd(dy, ctx.coord2d, flang.detach().contiguous(), dweight, e)
        ops.copy_slice(dw1, dweight[:, :e])
        dflang = ops.gemm_nn(d_img, wd[:, e:2 * e]) if ctx.needs_input_grad[1] else None
        return dcorr, dflang, None, dweight.view_as(weight), dgamma, dbeta, None, None, None


class CoAttentionPairs(torch.autograd.Function):
    """Inter-frame co-attention over consecutive frame pairs (model/DCNet_model.py:449-464).
    fv (N,H,W,C), unit norm over C, images 2p and 2p+1 form pair p.  Returns the concat
    [f | f_attn] (N,H,W,2C) in image order — the input of corr_conv."""

    @staticmethod
    def forward(ctx, fv, temperature: float):
        n, h, w, c = fv.shape
        hw = h * w
        cat = torch.empty((n, hw, 2 * c), dtype=torch.float32, device=fv.device)
        f = fv.view(n, hw, c)
        ops.copy_slice(f, cat[..., :c])
        E, rc = ops.coattn_fwd(f[0::2], f[1::2], cat[0::2, :, c:], cat[1::2, :, c:], temperature)
        out = cat.view(n, h, w, 2 * c)
        ctx.save_for_backward(fv, out, E, rc)
        ctx.temperature = temperature
        return out

    @staticmethod
    def backward(ctx, dcat):
        fv, out, E, rc = ctx.saved_tensors
        n, h, w, c = fv.shape
        hw = h * w
        f, cat = fv.view(n, hw, c), out.view(n, hw, 2 * c)
        dcat = dcat.contiguous().view(n, hw, 2 * c)
        df = torch.empty((n, hw, c), dtype=torch.float32, device=f.device)
        ops.copy_slice(dcat[..., :c], df)
        ops.coattn_bwd(f[0::2], f[1::2], dcat[0::2, :, c:], dcat[1::2, :, c:], cat[0::2, :, c:], cat[1::2, :, c:],
                       E, rc, df[0::2], df[1::2], True, ctx.temperature)
        return df.view(n, h, w, c), None


class CoAttentionCenter(torch.autograd.Function):
    """Centre-frame co-attention of the inference model (model/test_DCNet_model.py:247-282):
    f1 = centre frame, f2 = another frame of the clip; only f1_attn is consumed.
    clips (B,T,HW,C); returns [f1 | f1_attn] (B,HW,2C).
    Backward (the reference's train branch of this model, test_DCNet_model.py:480-483, reachable though its scripts never take it): the
    pair kernels of dcn_coattn_bwd with a zero gradient for the unused f2_attn; the result lands in frames ``ctr`` and ``idx`` of a
    clips-shaped gradient (autograd sums the T - 1 of them)."""

    @staticmethod
    def forward(ctx, clips, ctr: int, idx: int, temperature: float):
        b, t, hw, c = clips.shape
        cat = torch.empty((b, hw, 2 * c), dtype=torch.float32, device=clips.device)
        cat[..., :c].copy_(clips[:, ctr])        # batch-strided source: plain torch copy
        ctx.meta = (ctr, idx, temperature)
        if ctx.needs_input_grad[0]:
            o2 = torch.empty((b, hw, 2 * c), dtype=torch.float32, device=clips.device)[..., c:]      # f2_attn (strides of out1): only the backward's bookkeeping reads it
            E, rc = ops.coattn_fwd(clips[:, ctr], clips[:, idx], cat[..., c:], o2, temperature)
            ctx.save_for_backward(clips, cat, o2, E, rc)
        else:
            ops.coattn_fwd(clips[:, ctr], clips[:, idx], cat[..., c:], None, temperature)
        return cat

    @staticmethod
    def backward(ctx, g):
        clips, cat, o2, E, rc = ctx.saved_tensors
        ctr, idx, temperature = ctx.meta
        b, t, hw, c = clips.shape
        g = g.contiguous()
        d = torch.zeros_like(clips)
        d[:, ctr].copy_(g[..., :c])              # the pass-through half of the concat
        z2 = torch.zeros((b, hw, 2 * c), dtype=torch.float32, device=g.device)[..., c:]        # d f2_attn = 0, in the strides of d f1_attn
        ops.coattn_bwd(clips[:, ctr], clips[:, idx], g[..., c:], z2, cat[..., c:], o2, E, rc, d[:, ctr], d[:, idx], True, temperature)
        return d, None, None, None


class ToNCHW(torch.autograd.Function):
    """NHWC (first c channels) -> contiguous NCHW, for tensors the callers .view() (outbox)."""

    @staticmethod
    def forward(ctx, x, c: int):
        ctx.shape = tuple(x.shape)
        return ops.nhwc_to_nchw(x.contiguous(), c)

    @staticmethod
    def backward(ctx, g):
        n, h, w, ld = ctx.shape
        return ops.nchw_to_nhwc(g.contiguous(), ld), None


class LinearAct(torch.autograd.Function):
    """nn.Linear (+ optional ReLU) on the conv engine's GEMM forms: forward NT, input gradient NN, weight
    gradient TN.  x (M,K) with K % 32 == 0, weight (N,K) in nn.Linear's own layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool):
        x = x.contiguous()
        out = ops.gemm_nt(x, weight.detach(), None if bias is None else bias.detach(), ops.ACT_LEAKY if relu else ops.ACT_NONE)
        ctx.save_for_backward(x, weight, out if relu else x.new_empty(0))
        ctx.relu, ctx.has_bias = relu, bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        dz = dout.contiguous()
        if ctx.relu:
            dz = ops.act_bwd(out, dz, 0.0)
        dx = ops.gemm_nn(dz, weight.detach()) if ctx.needs_input_grad[0] else None
        dw = ops.gemm_tn(dz, x)
        db = ops.colsum(dz) if ctx.has_bias else None
        return dx, dw, db, None


class BatchNormRowsAct(torch.autograd.Function):
    """nn.BatchNorm1d (+ ReLU) over [rows][c] with the same HIP kernels as the 2-D case
    (model/DCNet_model.py:270,274)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, training: bool, relu: bool):
        x = x.contiguous()
        act = ops.ACT_LEAKY if relu else ops.ACT_NONE
        if training:
            mi = ops.bn_finalize(ops.channel_stats(x), x.shape[0], gamma.detach(), beta.detach(), bn.eps, bn.momentum,
                                 bn.running_mean, bn.running_var)
            ops.bump_batches(bn)
            out = ops.scale_act(x, mi[2], mi[3], act, 0.0)
            ctx.save_for_backward(x, mi, gamma, beta)
        else:
            ss = ops.bn_fold(gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, bn.eps)
            out = ops.scale_act(x, ss[0], ss[1], act, 0.0)
            ctx.save_for_backward(out, ss, gamma, beta)
        ctx.training, ctx.relu = training, relu
        return out

    @staticmethod
    def backward(ctx, dout):
        a, aux, gamma, beta = ctx.saved_tensors
        gamma, beta = gamma.detach(), beta.detach()
        dout = dout.contiguous()
        act = ops.ACT_LEAKY if ctx.relu else ops.ACT_NONE
        if ctx.training:
            dx, dgamma, dbeta = ops.bn_act_bwd(a, dout, aux[0], aux[1], gamma, beta, act, 0.0)
        else:
            dz = ops.act_bwd(a, dout, 0.0) if ctx.relu else dout
            dx = dz * aux[0]
            dbeta = dz.sum(0)
            gs = torch.where(gamma == 0, torch.ones_like(gamma), gamma)
            dgamma = (dz * (a - beta) / gs).sum(0)
        return dx, dgamma, dbeta, None, None, None


class Embedding(torch.autograd.Function):
    """nn.Embedding (model/DCNet_model.py:168): row gather; the backward sums the rows of each vocabulary entry in token order
    (deterministic, no sort / no atomics)."""

    @staticmethod
    def forward(ctx, ids, table):
        ids = ids.contiguous()
        ctx.save_for_backward(ids)
        ctx.vocab = table.shape[0]
        return ops.embedding_fwd(ids, table.detach().contiguous())

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        return None, ops.embedding_bwd(ids, dout.contiguous(), ctx.vocab)


class BiLSTM(torch.autograd.Function):
    """One-layer bidirectional LSTM over (N,L,I) with per-row lengths (packed-sequence semantics:
    model/DCNet_model.py:172-183): the input projections of all steps as one GEMM per direction, then the whole
    recurrence — 2 x L dependent steps — as ONE persistent launch (csrc/lstm.hip); the backward mirrors it and
    finishes with the weight / input gradients as GEMMs over all (row, time) pairs.  Returns (N,L,2H).
    Parameter order: w_ih, w_hh, b_ih, b_hh for the forward direction, then the reverse direction."""

    @staticmethod
    def forward(ctx, x, lengths, *params):
        n, L, I = x.shape
        H = params[1].shape[1]
        x2d = x.contiguous().view(n * L, I)
        det = [p.detach().contiguous() for p in params]
        xg = torch.empty((2, n, L, 4 * H), dtype=torch.float32, device=x.device)
        for d in range(2):
            ops.gemm_nt(x2d, det[4 * d], det[4 * d + 2], out=xg[d].view(n * L, 4 * H))
        lens = lengths.contiguous()
        out, hprev, cprev, acts = ops.bilstm_fwd(xg, det[1], det[5], det[3], det[7], lens)
        ctx.save_for_backward(x2d, lens, acts, hprev, cprev, *params)
        ctx.dims = (n, L, I, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2d, lens, acts, hprev, cprev, *params = ctx.saved_tensors
        n, L, I, H = ctx.dims
        det = [p.detach().contiguous() for p in params]
        dxg = ops.bilstm_bwd(dout.contiguous(), det[1], det[5], acts, cprev, lens)
        dx = torch.empty((n * L, I), dtype=torch.float32, device=x2d.device)
        grads = []
        for d in range(2):
            g2 = dxg[d].view(n * L, 4 * H)
            dw_ih = ops.gemm_tn(g2, x2d)
            dw_hh = ops.gemm_tn(g2, hprev[d].view(n * L, H))                # sum over (row, time) of dgates^T . h_before
            ops.gemm_nn(g2, det[4 * d], out=dx, accumulate=(d == 1))
            db = ops.rows_sum(g2)
            grads += [dw_ih, dw_hh, db, db]
        return (dx.view(n, L, I), None) + tuple(grads)


class LocModule(torch.autograd.Function):
    """loc[n,i] = < normalize(relu(E_i . Mp_n + bp)), q_n >  — the fused rank-8 location module
    (csrc/locmod.hip; reference model/DCNet_model.py:581-594).  E (P,8), Mp (N,8,512), bp (512), q (N,512)."""

    @staticmethod
    def forward(ctx, E, Mp, bp, q):
        E, Mp, bp, q = E.contiguous(), Mp.contiguous(), bp.contiguous(), q.contiguous()
        ctx.save_for_backward(E, Mp, bp, q)
        return ops.locmod_fwd(E, Mp, bp, q)

    @staticmethod
    def backward(ctx, dloc):
        E, Mp, bp, q = ctx.saved_tensors
        dE_part, dsum = ops.locmod_bwd(E, Mp, bp, q, dloc.contiguous())
        return dE_part.sum(0), dsum[:, :8], dsum[:, 8].sum(0), dsum[:, 9]
