"""DCNet grounding model on the MI355X kernel library.

Drop-in for the reference's ``model/DCNet_model.py::grounding_model`` (train model, T = 2 frame
pairs) and ``model/test_DCNet_model.py::grounding_model`` (inference model, ``n_frame``): same
constructor arguments, same ``forward(image, word_id, word_mask[, n_frame])`` signature, same
return tuples, same sub-module names and therefore the same 597 state_dict keys.

Differences that are deliberate:
  * ``img_size`` / ``query_len`` / ``weights_path`` / ``config_path`` keyword arguments replace the
    literals 1344 (= P at 256x256), 20 and the cwd-relative file names (F3, F4, F6);
  * the dead YOLO heads are not executed (F7) — their parameters stay in the state_dict;
  * ``forward`` accepts an optional 4th argument ``n_frame`` and then follows the inference model's
    semantics (F2);
  * the location module uses the rank-8 identity of SURVEY.md K13 (no PxP tensor);
  * negative sampling runs natively on the host with Python's own MT19937 stream (bit-exact with
    ``random.sample``; see csrc/sampling.cpp).
"""
from __future__ import annotations

import os
import random
import threading
from collections import OrderedDict
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import losses as losses_mod
from . import ops
from .darknet import Darknet
from .functions import (BatchNormRowsAct, BiLSTM, CoAttentionCenter, CoAttentionPairs, ConvBias, ConvBNAct, FusionConvBNAct,
                        L2Norm, LinearAct, LocModule, NormScore, ToNCHW)
from .lib import lib


class ConvBatchNormReLU(nn.Sequential):
    """Parameter container with the reference's layout (model/darknet.py:118-156): ``conv`` (no bias),
    ``bn`` (eps 1e-5, momentum 0.999), ``relu``.  Executed through functions.ConvBNAct on NHWC maps."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, leaky=False, relu=True):
        super().__init__()
        if stride != 1 or dilation != 1 or padding != (kernel_size - 1) // 2:
            raise NotImplementedError("DCNet head blocks are stride 1, dilation 1, 'same' padding")
        self.add_module("conv", nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, bias=False))
        self.add_module("bn", nn.BatchNorm2d(out_channels, eps=1e-5, momentum=0.999, affine=True))
        self.slope = 0.1 if leaky else 0.0
        if leaky:
            self.add_module("relu", nn.LeakyReLU(0.1))
        elif relu:
            self.add_module("relu", nn.ReLU())
        else:
            raise NotImplementedError("relu=False is not used by DCNet")

    def forward(self, x_nhwc):
        return ConvBNAct.apply(x_nhwc, self.conv.weight, self.bn.weight, self.bn.bias, self.bn,
                               self.conv.kernel_size[0], self.training, self.slope)


class RNNEncoder(nn.Module):
    """model/DCNet_model.py:124-188.  Every row is processed to its own length exactly like the
    packed BiLSTM of the reference; with the reference's tokenizer all lengths equal L (F4)."""

    def __init__(self, vocab_size, word_embedding_size, word_vec_size, hidden_size, bidirectional=False,
                 input_dropout_p=0, dropout_p=0, n_layers=1, rnn_type="lstm", variable_lengths=True):
        super().__init__()
        self.variable_lengths = variable_lengths
        self.embedding = nn.Embedding(vocab_size, word_embedding_size)
        self.input_dropout = nn.Dropout(input_dropout_p)
        self.mlp = nn.Sequential(nn.Linear(word_embedding_size, word_vec_size), nn.ReLU())
        self.rnn_type = rnn_type
        self.rnn = getattr(nn, rnn_type.upper())(word_vec_size, hidden_size, n_layers, batch_first=True,
                                                 bidirectional=bidirectional, dropout=dropout_p)
        self.num_dirs = 2 if bidirectional else 1

    def forward(self, input_labels):
        # No host synchronisation: lengths stay on the device and the LSTM kernels mask finished rows
        # themselves.  The reference trims the batch to its longest row first (model/DCNet_model.py:474-475);
        # computing the padded steps instead changes nothing that is consumed (padded outputs are zero,
        # PhraseAttention renormalises over the valid positions, the sentence vector is read at len-1).
        lengths = (input_labels != 0).sum(1)
        n, L = input_labels.shape
        emb = self.input_dropout(self.embedding(input_labels))                              # :168-169
        lin = self.mlp[0]
        embedded = LinearAct.apply(emb.view(n * L, -1), lin.weight, lin.bias, True).view(n, L, -1)     # :170
        r = self.rnn
        output = BiLSTM.apply(embedded, lengths,
                              r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0,
                              r.weight_ih_l0_reverse, r.weight_hh_l0_reverse, r.bias_ih_l0_reverse, r.bias_hh_l0_reverse)
        embedded = embedded * (torch.arange(L, device=embedded.device)[None, :]
                               < lengths[:, None]).unsqueeze(2).to(embedded.dtype)              # :178 pad_packed zeros
        sent = output[torch.arange(output.size(0), device=output.device), lengths - 1]
        return sent, output, embedded


class PhraseAttention(nn.Module):
    """model/DCNet_model.py:190-219."""

    def __init__(self, input_dim):
        super().__init__()
        self.fc = nn.Linear(input_dim, 1)

    def forward(self, context, embedded, input_labels):
        attn = F.softmax(self.fc(context).squeeze(2), dim=1)
        attn = attn * (input_labels != 0).float()
        attn = attn / attn.sum(1, keepdim=True)
        return attn, torch.bmm(attn.unsqueeze(1), embedded).squeeze(1)


def generate_coord_nhwc(height: int, width: int, device) -> torch.Tensor:
    """The 8-channel coordinate map of model/DCNet_model.py:23-39 as an (H,W,8) NHWC constant
    (it is identical for every image).  ``xv`` indexes rows, as in the reference."""
    xv, yv = torch.meshgrid([torch.arange(0, height), torch.arange(0, width)], indexing="ij")
    xv_min = (xv.float() * 2 - width) / width
    yv_min = (yv.float() * 2 - height) / height
    xv_max = ((xv + 1).float() * 2 - width) / width
    yv_max = ((yv + 1).float() * 2 - height) / height
    xv_ctr = (xv_min + xv_max) / 2
    yv_ctr = (yv_min + yv_max) / 2
    hmap = torch.ones(height, width) * (1. / height)
    wmap = torch.ones(height, width) * (1. / width)
    return torch.stack([xv_min, yv_min, xv_max, yv_max, xv_ctr, yv_ctr, hmap, wmap], dim=2).to(device)


def generate_coord(batch, height, width, device="cuda"):
    """NCHW form with the reference's signature (model/DCNet_model.py:23)."""
    return generate_coord_nhwc(height, width, device).permute(2, 0, 1).unsqueeze(0).repeat(batch, 1, 1, 1)


def _mt_state():
    st = random.getstate()
    return st, np.array(st[1], dtype=np.uint32)


def _mt_restore(st, arr):
    random.setstate((st[0], tuple(int(v) for v in arr), st[2]))


class grounding_model(nn.Module):
    def __init__(self, corpus=None, emb_size=256, jemb_drop_out=0.1, bert_model="bert-base-uncased",
                 coordmap=True, leaky=False, dataset=None, light=False,
                 img_size: int = 256, query_len: int = 20, config_path: str = "./model/yolov3.cfg",
                 weights_path: Optional[str] = "./saved_models/yolov3.weights"):
        super().__init__()
        if corpus is None:
            raise NotImplementedError("corpus=None selects the BERT text encoder, which is outside the LSTM hot path "
                                      "(train_DCNet.py is run with --lstm); pass the dataset corpus")
        if light:
            raise NotImplementedError("light=True (1-layer fcn_emb) is not used by the released scripts")
        if not coordmap:
            raise NotImplementedError("coordmap=False is not used by the released scripts")
        self.coordmap, self.light, self.lstm, self.emb_size = coordmap, light, True, emb_size
        self.img_size, self.query_len = img_size, query_len
        self.grids = [img_size // 32, img_size // 16, img_size // 8]
        self.num_pos = sum(g * g for g in self.grids)          # 1344 at 256x256 (model/DCNet_model.py:259,584)
        self.textdim, self.embdim = 1024, 512
        ## Visual model
        self.visumodel = Darknet(config_path=config_path)
        if weights_path and os.path.exists(weights_path):
            self.visumodel.load_weights(weights_path)
        ## Text model
        self.textmodel = RNNEncoder(vocab_size=len(corpus), word_embedding_size=self.embdim,
                                    word_vec_size=self.textdim // 2, hidden_size=self.textdim // 2,
                                    bidirectional=True, input_dropout_p=0.2, variable_lengths=True)
        self.temperature = 10.
        self.sub_attn = PhraseAttention(self.textdim)
        self.loc_embedding = nn.Sequential(nn.Linear(8, 8), nn.BatchNorm1d(8), nn.ReLU())
        self.loc_text_embedding = nn.Sequential(nn.Linear(self.num_pos, self.embdim), nn.BatchNorm1d(self.embdim), nn.ReLU())
        self.loc_attn = PhraseAttention(self.textdim)
        self.mapping_visu = nn.Sequential(OrderedDict([
            ("0", ConvBatchNormReLU(1024, emb_size, 1, 1, 0, 1, leaky=leaky)),
            ("1", ConvBatchNormReLU(512, emb_size, 1, 1, 0, 1, leaky=leaky)),
            ("2", ConvBatchNormReLU(256, emb_size, 1, 1, 0, 1, leaky=leaky))]))
        self.mapping_lang = nn.Sequential(
            nn.Linear(self.textdim, emb_size), nn.BatchNorm1d(emb_size), nn.ReLU(), nn.Dropout(jemb_drop_out),
            nn.Linear(emb_size, emb_size), nn.BatchNorm1d(emb_size), nn.ReLU())
        self.corr_conv = nn.Sequential(OrderedDict([
            (str(i), nn.Sequential(ConvBatchNormReLU(emb_size * 2, emb_size, 1, 1, 0, 1, leaky=leaky))) for i in range(3)]))
        self.feature_map = nn.Sequential(
            nn.Conv1d(in_channels=query_len, out_channels=query_len, stride=1, kernel_size=3, padding=1, bias=True),
            nn.Softmax(dim=1))
        embin_size = emb_size * 2 + 8
        self.fcn_emb = nn.Sequential(OrderedDict([
            (str(i), nn.Sequential(ConvBatchNormReLU(embin_size, emb_size, 1, 1, 0, 1, leaky=leaky),
                                   ConvBatchNormReLU(emb_size, emb_size, 3, 1, 1, 1, leaky=leaky),
                                   ConvBatchNormReLU(emb_size, emb_size, 1, 1, 0, 1, leaky=leaky))) for i in range(3)]))
        self.fcn_out = nn.Sequential(OrderedDict([
            (str(i), nn.Sequential(ConvBatchNormReLU(emb_size, emb_size // 2, 1, 1, 0, 1, leaky=leaky),
                                   nn.Conv2d(emb_size // 2, 3 * 5, kernel_size=1))) for i in range(3)]))
        self._coord_cache = {}
        self._pinned = {}
        self._pin_event = None
        self._streams = {}
        # run the three per-scale head branches on separate streams: -1 % step time in an in-process A/B, but
        # the HBM-bound scoring kernels then share the memory system with the other scales' GEMMs; off by default
        self.scale_streams = False
        # the two sampling heads (K9, K14) on their own stream under the head convs of scales 1 and 2 (forward()), and
        # their contrastive losses on that stream too (losses.total_loss), so that their backward overlaps as well
        self.sampling_stream = True
        # discrete choices of the last forward (top-k / arg-max indices and the sampled negatives):
        # exposed so that parity tests can replay them through the oracle (near-tie orderings differ
        # between fp32 implementations) and so that a caller can log what was sampled
        self.last_choices = {}

    # ------------------------------------------------------------------------------------------
    def _side_stream(self, device, name: str = "lang"):
        key = (str(device), name)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def _coord(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._coord_cache:
            self._coord_cache[key] = generate_coord_nhwc(h, w, device)
        return self._coord_cache[key]

    def _language(self, word_id):
        raw_flang, context, embedded = self.textmodel(word_id)                  # DCNet_model.py:474-476 (untrimmed, see RNNEncoder)
        ml = self.mapping_lang                                                  # Linear, BN1d, ReLU, Dropout, Linear, BN1d, ReLU
        z = LinearAct.apply(raw_flang, ml[0].weight, ml[0].bias, False)
        z = BatchNormRowsAct.apply(z, ml[1].weight, ml[1].bias, ml[1], self.training, True)
        z = ml[3](z)
        z = LinearAct.apply(z, ml[4].weight, ml[4].bias, False)
        z = BatchNormRowsAct.apply(z, ml[5].weight, ml[5].bias, ml[5], self.training, True)
        flang = F.normalize(z, p=2, dim=1)                                      # :485-487
        return word_id, flang, context, embedded

    def _fusion_head(self, s: int, corr, flang):
        """fcn_emb[s] + fcn_out[s] on one scale: corr (B,H,W,E) -> outbox logits (B,H,W,15)  (:491-506)."""
        h, w = corr.shape[1], corr.shape[2]
        blk0 = self.fcn_emb[s][0]                                                # [corr | tile(flang) | coord] -> 1x1
        z = FusionConvBNAct.apply(corr.contiguous(), flang, self._coord(h, w, corr.device), blk0.conv.weight,
                                  blk0.bn.weight, blk0.bn.bias, blk0.bn, self.training)
        z = self.fcn_emb[s][1](z)
        z = self.fcn_emb[s][2](z)
        z = self.fcn_out[s][0](z)
        last = self.fcn_out[s][1]
        return ConvBias.apply(z, last.weight, last.bias)[..., :15]

    def _scale_pairs(self, s: int, raw_s, flang, flang_attn):
        """Everything of scale s that depends only on its backbone tap (pair semantics): mapping + norm
        (:356-359), co-attention + corr_conv (:449-468), normalise + sim (:469,530-535), fusion head."""
        fv = L2Norm.apply(self.mapping_visu[s](raw_s))
        corr_raw = self.corr_conv[s][0](CoAttentionPairs.apply(fv, self.temperature))
        corr, sim = NormScore.apply(corr_raw, flang_attn)
        return fv, corr, sim, self._fusion_head(s, corr, flang)

    def _scale_nframe(self, s: int, raw_s, flang, flang_attn, B: int, n_frame: int):
        """Scale s of the inference model: centre frame vs every other frame, mean of the normalised
        correspondence features (model/test_DCNet_model.py:299-332), then the shared head."""
        fv = L2Norm.apply(self.mapping_visu[s](raw_s))
        _, h, w, e = fv.shape
        clips = fv.view(B, n_frame, h * w, e)
        ctr, acc = n_frame // 2, None                                            # :303
        for idx in range(n_frame):                                               # :312-320
            if idx == ctr:
                continue
            cat = CoAttentionCenter.apply(clips, ctr, idx, self.temperature).view(B, h, w, 2 * e)
            z = L2Norm.apply(self.corr_conv[s][0](cat))                          # :277-280
            acc = z if acc is None else acc + z
        corr = acc / (n_frame - 1)                                               # :324-332
        sim = torch.sum(corr * flang_attn.view(B, 1, 1, -1), dim=3)
        return fv, corr, sim, self._fusion_head(s, corr, flang)

    def _run_scales(self, branch, raws, main):
        """Run the three per-scale branches on their own streams (finest scale first): the 13x13 and 26x26
        branches are chains of small grids that fit under the 52x52 branch's kernels.  Autograd replays
        each branch's backward on the same stream."""
        dev = raws[0].device
        res = [None] * 3
        if not self.scale_streams:
            return [branch(s_, raws[s_]) for s_ in range(3)]
        for s_ in (2, 1, 0):
            st = self._side_stream(dev, f"scale{s_}")
            st.wait_stream(main)
            with torch.cuda.stream(st):
                res[s_] = branch(s_, raws[s_])
            raws[s_].record_stream(st)
        for s_ in range(3):
            main.wait_stream(self._side_stream(dev, f"scale{s_}"))
            for t_ in res[s_]:
                t_.record_stream(main)
        return res

    def _head(self, corr_feat, sim_score, outbox, word_id, flang, context, embedded, flang_attn):
        """The cross-scale tail: objectness x similarity, location module, confidence modulation
        (:545-621).  Inputs are NHWC per-scale tensors."""
        B = flang.shape[0]
        dev = flang.device
        conf = [ob.reshape(B, ob.shape[1], ob.shape[2], 3, 5)[..., 4] for ob in outbox]   # (B,H,W,3)
        only_obj = [c.mean(dim=3) for c in conf]                                 # :551
        obj_score = [o * s for o, s in zip(only_obj, sim_score)]                 # :550

        _, flang_loc = self.loc_attn(context, embedded, word_id)                 # :556
        flang_loc = F.normalize(flang_loc, p=2, dim=1)
        # ---- location module, rank-8 form (reference :559-597 builds a PxP tensor) ----
        coord_map = torch.cat([self._coord(c.shape[1], c.shape[2], dev).reshape(-1, 8) for c in corr_feat], dim=0)  # (P,8)
        P = coord_map.shape[0]
        obj_map = F.normalize(torch.cat([o.reshape(B, -1) for o in obj_score], dim=1), p=2, dim=1)       # :566-569
        lin, bn = self.loc_embedding[0], self.loc_embedding[1]
        ce = F.linear(coord_map, lin.weight, lin.bias)                           # identical rows for every image
        if self.training:
            # BN over the B*P rows == BN over the P distinct rows (each repeated B times); the
            # unbiased running_var uses the true row count B*P
            mean = ce.mean(0); var = ce.var(0, unbiased=False)
            with torch.no_grad():
                cnt = B * P
                bn.running_mean.mul_(1 - bn.momentum).add_(bn.momentum * mean)
                bn.running_var.mul_(1 - bn.momentum).add_(bn.momentum * var * cnt / (cnt - 1))
                bn.num_batches_tracked += 1
            ce = (ce - mean) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias
        else:
            ce = F.batch_norm(ce, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        E8 = F.normalize(F.relu(ce), p=2, dim=1)                                 # (P,8)   :578
        lt, bn2 = self.loc_text_embedding[0], self.loc_text_embedding[1]
        # rel[n,i,:] = sum_j <E_i,E_j> obj[n,j] W[:,j] + b  =  E_i . (E^T diag(obj_n) W^T) + b = E_i . M_n + b
        M = torch.matmul((E8.t().unsqueeze(0) * obj_map.unsqueeze(1)), lt.weight.t())     # (B,8,512)   :581-585
        # BatchNorm1d over the B*P rows of rel, from the 8x8 moments of E (no (B,P,512) tensor): with
        # s1 = sum_i E_i and S2 = sum_i E_i E_i^T,  sum(rel) = sum_n s1.M_n + BP b,
        # sum(rel^2) = sum_n M_n^T S2 M_n + 2 b sum_n s1.M_n + BP b^2   (fp64 on these tiny tensors)
        cnt = float(B * P)
        if self.training:
            Ed, Md, bd = E8.double(), M.double(), lt.bias.double()
            # (broadcast products, not matmul/einsum: rocBLAS runs these 8-wide fp64 shapes on a 128x128 DGEMM
            #  tile — one of them took 3.5 ms — while the tensors here are a few MB)
            s1 = Ed.sum(0); S2 = (Ed.unsqueeze(2) * Ed.unsqueeze(1)).sum(0)                  # (8,), (8,8)
            t1 = (s1.view(1, 8, 1) * Md).sum((0, 1))
            mean = t1 / cnt + bd
            S2M = (S2.view(1, 8, 8, 1) * Md.unsqueeze(1)).sum(2)                              # (B,8,512)
            ex2 = ((Md * S2M).sum((0, 1)) + 2 * bd * t1) / cnt + bd * bd                      # M_n^T S2 M_n per channel
            var = torch.clamp(ex2 - mean * mean, min=0)
            with torch.no_grad():
                bn2.running_mean.mul_(1 - bn2.momentum).add_(bn2.momentum * mean.float())
                bn2.running_var.mul_(1 - bn2.momentum).add_(bn2.momentum * (var * cnt / (cnt - 1)).float())
                bn2.num_batches_tracked += 1
            scale = bn2.weight.double() * torch.rsqrt(var + bn2.eps)
            shift = bn2.bias.double() - mean * scale
            scale, shift = scale.float(), shift.float()
        else:
            scale = bn2.weight * torch.rsqrt(bn2.running_var + bn2.eps)
            shift = bn2.bias - bn2.running_mean * scale
        # relu(bn(rel)) -> normalize over channels -> <., flang_loc>, fused on the device   :585-594
        loc_map = LocModule.apply(E8, M * scale, lt.bias * scale + shift, flang_loc)       # (B,P)
        mn = loc_map.min(dim=1, keepdim=True)[0]; mx = loc_map.max(dim=1, keepdim=True)[0]
        loc_map = (loc_map - mn) / (mx - mn + 1e-6)                              # :597
        loc_score, st = [], 0
        for c in corr_feat:
            h, w = c.shape[1], c.shape[2]
            loc_score.append(loc_map[:, st:st + h * w].reshape(B, h, w)); st += h * w
        final = []
        for s in range(3):                                                       # :612-621
            ob = outbox[s].reshape(B, outbox[s].shape[1], outbox[s].shape[2], 3, 5)
            c4 = ob[..., 4] * (sim_score[s] * loc_score[s]).unsqueeze(3)
            ob = torch.cat([ob[..., :4], c4.unsqueeze(4)], dim=4).reshape(B, ob.shape[1], ob.shape[2], 15)
            final.append(ToNCHW.apply(ob, 15))
        return final, loc_score, only_obj

    # ------------------------------------------------------------------------------------------
    def _presample_start(self, n, g0, top_k=30, neg_n=10, neg_c=5):
        """Draw the negatives of both correspondence heads from Python's global MT19937 stream, in the
        reference's order (K9 then K14), natively and on a worker thread (ctypes drops the GIL inside the
        call): the draws depend only on shapes, so they overlap with queueing and running the backbone."""
        hw = g0 * g0
        # random.sample raises ValueError for these in the reference (model/DCNet_model.py:412,88); so do we, up front
        if n < 2 or hw - 1 < neg_n or hw - 1 < neg_c or hw * hw < top_k:
            raise ValueError(f"correspondence sampling needs >= 2 images and a coarsest grid of more than {max(neg_n, neg_c)} "
                             f"cells with >= {top_k} cell pairs (got {n} images, {g0}x{g0})")
        key = (n, hw, top_k, neg_n, neg_c)
        if key not in self._pinned:
            # page-locked staging so the upload is a true async copy (a pageable H2D would block the host
            # until the queued backbone kernels drain).  Reuse is safe: the language branch of the next
            # forward syncs the stream before these buffers are written again.
            self._pinned[key] = (torch.empty((n // 2, top_k, neg_n), dtype=torch.int64).pin_memory(),
                                 torch.empty((n, hw, neg_c), dtype=torch.int64).pin_memory())
        k9, k14 = self._pinned[key]
        if self._pin_event is not None:
            self._pin_event.synchronize()        # the previous forward's upload of these buffers has completed
        st, arr = _mt_state()
        L = lib()
        box = {"err": None, "shape": (n, top_k, hw, neg_n, neg_c)}

        def work():
            try:
                L.mt_sample_interframe(arr.ctypes.data, 0, n // 2, top_k, hw, neg_n, k9.data_ptr())
                L.mt_sample_crossmodal(arr.ctypes.data, n, hw, neg_c, k14.data_ptr())
            except BaseException as e:       # re-raised on the caller's thread by _presample_join
                box["err"] = e

        th = threading.Thread(target=work, daemon=True)
        th.start()
        return th, st, arr, k9, k14, box

    def _presample_join(self, handle, device):
        th, st, arr, k9, k14, box = handle
        th.join()
        if box["err"] is not None:
            raise box["err"]
        if random.getstate() != st:
            # someone drew from the global stream while the worker was running (another thread: forward itself draws
            # nothing else).  The worker started from a stale state: redo the draws from the current one, synchronously.
            st, arr = _mt_state()
            n, top_k, hw, neg_n, neg_c = box["shape"]
            L = lib()
            L.mt_sample_interframe(arr.ctypes.data, 0, n // 2, top_k, hw, neg_n, k9.data_ptr())
            L.mt_sample_crossmodal(arr.ctypes.data, n, hw, neg_c, k14.data_ptr())
        _mt_restore(st, arr)
        out = {"k9": k9.to(device, non_blocking=True), "k14": k14.to(device, non_blocking=True)}
        ev = torch.cuda.Event(); ev.record()
        return out, ev

    def _interframe_sampling(self, fv0, presampled, top_k=30, neg_n=10):
        """model/DCNet_model.py:381-430 on the NHWC scale-0 map (N,g,g,E)."""
        n, g, _, e = fv0.shape
        hw = g * g
        f = fv0.reshape(n // 2, 2, hw, e)
        p1, p2 = f[:, 0], f[:, 1]
        cmap = torch.bmm(p1, p2.transpose(1, 2)).flatten(1)                      # :390  [i*hw + j]
        _, index = cmap.topk(top_k, dim=1, largest=True, sorted=True)            # :395
        qi, ki = index // hw, index % hw                                         # :407,409
        raw = presampled["k9"]                                                   # list positions, drawn on the host
        ni = raw + (raw >= ki.unsqueeze(2)).long()                               # skip the removed element (:411-413)
        self.last_choices["k9_index"] = index.detach()
        self.last_choices["k9_neg"] = ni
        # one batched gather per output; the reference's lists are zero-copy unbinds of them
        ar = torch.arange(n // 2, device=fv0.device)
        frame = p1[ar.unsqueeze(1), qi]                                          # (b,top_k,E)
        corr = p2[ar.unsqueeze(1), ki]
        negf = p2[ar.view(-1, 1, 1), ni]                                         # (b,top_k,neg_n,E)
        return list(frame.unbind(1)), list(corr.unbind(1)), list(negf.unbind(1))

    def _crossmodal(self, fv0, context, presampled, neg_n=5):
        """model/DCNet_model.py:625-637 + Crossmodal_corrspondence :41-112."""
        n, g, _, e = fv0.shape
        hw = g * g
        v = fv0.reshape(n, hw, e)
        vit = F.normalize(v, dim=1)                                              # over positions (:629)
        lag = F.normalize(context[:, :, 0::2], dim=1)                            # interpolate(0.5) + over L (:631-632)
        lv = torch.bmm(lag, vit.transpose(1, 2))                                 # (N,L,HW0)  :634
        lv = self.feature_map(lv)                                                # :635
        cols = lv.argmax(dim=1)                                                  # top-1 word per position (:48)
        ni = presampled["k14"]
        self.last_choices["k14_cols"] = cols.detach()
        self.last_choices["k14_neg"] = ni
        ar = torch.arange(n, device=fv0.device)
        lag_pos = lag[ar.unsqueeze(1), cols].unsqueeze(2)                        # (N,HW0,1,E)
        neg_cross = vit[n - 1][ni]                                               # (N,HW0,neg_n,E)
        return list(vit.unbind(1)), list(lag_pos.unbind(1)), list(neg_cross.unbind(1))

    # ------------------------------------------------------------------------------------------
    def forward(self, image, word_id, word_mask=None, n_frame: Optional[int] = None):
        if not image.is_cuda:
            raise RuntimeError("dcnet_amd.grounding_model runs on an MI355X only (HIP kernels, no CPU path)")
        if n_frame is not None:
            return self._forward_nframe(image, word_id, n_frame)
        N = image.size(0)
        if N % 2:
            raise ValueError("the training model consumes frame pairs: batch must be even (model/DCNet_model.py:365)")
        # The forward issues no host synchronisation at all (lengths are handled on the device), so the
        # host can queue step k+1 while the GPU still runs step k.  The language branch (a chain of ~100
        # latency-bound small kernels) goes on a side stream: it runs under the backbone, and autograd
        # replays its backward on the same side stream under the backbone's backward.
        main = torch.cuda.current_stream()
        side = self._side_stream(image.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            word_id, flang, context, embedded = self._language(word_id)
        handle = self._presample_start(N, image.shape[-1] // 32)                 # worker thread, under the backbone
        raw = self.visumodel.forward_nhwc(image)                                 # :344  (queued asynchronously)
        main.wait_stream(side)
        for t_ in (flang, context, embedded):
            t_.record_stream(main)
        _, flang_attn = self.sub_attn(context, embedded, word_id)                # :525
        flang_attn = F.normalize(flang_attn, p=2, dim=1)                         # :526
        if self.sampling_stream and not self.scale_streams:
            # The two correspondence-sampling heads read only the scale-0 features: a few hundred tiny launches (top-k,
            # gathers, normalisations) that otherwise sit between the last head conv and the first backward kernel with
            # nothing to overlap.  They go on their own stream as soon as scale 0 is queued and run under the head convs
            # of scales 1 and 2; autograd replays their backward on the same stream, beside the heads' backward.
            r0 = self._scale_pairs(0, raw[0], flang, flang_attn)
            presampled, self._pin_event = self._presample_join(handle, image.device)
            samp = self._side_stream(image.device, "samp")
            samp.wait_stream(main)
            with torch.cuda.stream(samp):
                frame_feature, corrspendence_feature, neg_feature = self._interframe_sampling(r0[0], presampled)   # :381-430
                vit_posit, lag_posit, neg_cross = self._crossmodal(r0[0], context, presampled)                      # :625-637
            for t_ in (r0[0], context, presampled["k9"], presampled["k14"]):
                t_.record_stream(samp)
            res = [r0, self._scale_pairs(1, raw[1], flang, flang_attn), self._scale_pairs(2, raw[2], flang, flang_attn)]
            fv = [r[0] for r in res]; corr_feat = [r[1] for r in res]; sim = [r[2] for r in res]
            outbox, loc, only_obj = self._head(corr_feat, sim, [r[3] for r in res], word_id, flang, context, embedded, flang_attn)
            flang_attn = flang_attn.view(N, -1, 1, 1)
            main.wait_stream(samp)               # callers read the sampled lists on the current stream
            for t_ in frame_feature + corrspendence_feature + neg_feature + vit_posit + lag_posit + neg_cross:
                t_.record_stream(main)
            losses_mod.CONTRASTIVE_STREAM = samp if (self.training and torch.is_grad_enabled()) else None
        else:
            losses_mod.CONTRASTIVE_STREAM = None
            res = self._run_scales(lambda s_, r_: self._scale_pairs(s_, r_, flang, flang_attn), raw, main)
            fv = [r[0] for r in res]; corr_feat = [r[1] for r in res]; sim = [r[2] for r in res]
            presampled, self._pin_event = self._presample_join(handle, image.device)
            frame_feature, corrspendence_feature, neg_feature = self._interframe_sampling(fv[0], presampled)   # :381-430
            outbox, loc, only_obj = self._head(corr_feat, sim, [r[3] for r in res], word_id, flang, context, embedded, flang_attn)
            flang_attn = flang_attn.view(N, -1, 1, 1)
            vit_posit, lag_posit, neg_cross = self._crossmodal(fv[0], context, presampled)   # :625-637 (runs in eval too)
        if self.training:
            return (outbox, sim, loc, [c.permute(0, 3, 1, 2) for c in corr_feat], flang_attn,
                    frame_feature, corrspendence_feature, neg_feature, vit_posit, lag_posit, neg_cross)
        return outbox, sim, loc, only_obj

    def _forward_nframe(self, image, word_id, n_frame: int):
        """model/test_DCNet_model.py:284-483."""
        if image.size(0) % n_frame:
            raise ValueError("batch must be a multiple of n_frame (model/test_DCNet_model.py:287)")
        B = image.size(0) // n_frame
        main = torch.cuda.current_stream()
        side = self._side_stream(image.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            word_id, flang, context, embedded = self._language(word_id)
        raw = self.visumodel.forward_nhwc(image)
        main.wait_stream(side)
        for t_ in (flang, context, embedded):
            t_.record_stream(main)
        _, flang_attn = self.sub_attn(context, embedded, word_id)
        flang_attn = F.normalize(flang_attn, p=2, dim=1)
        res = self._run_scales(lambda s_, r_: self._scale_nframe(s_, r_, flang, flang_attn, B, n_frame), raw, main)
        corr_feat = [r[1] for r in res]; sim = [r[2] for r in res]
        outbox, loc, only_obj = self._head(corr_feat, sim, [r[3] for r in res], word_id, flang, context, embedded, flang_attn)
        flang_attn = flang_attn.view(B, -1, 1, 1)
        corr_nchw = [c.permute(0, 3, 1, 2) for c in corr_feat]
        if self.training:
            return outbox, sim, loc, corr_nchw, flang_attn
        return outbox, sim, loc, corr_nchw, only_obj
