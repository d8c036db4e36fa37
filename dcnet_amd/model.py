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
import ctypes
import random
import threading
import time
from collections import OrderedDict
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .darknet import Darknet
from .functions import (BatchNormRowsAct, BiLSTM, CoAttentionCenter, CoAttentionPairs, ConvBias, ConvBNAct, CrossModalSample,
                        Embedding, FusionConvBNAct, HeadTail, InterframeSample, L2Norm, LinearAct, NormAccumulate, NormScore, PhraseAttn, RowDot)
from .lib import lib


class OutputList(list):
    """A plain list (what the reference returns and its callers index / overwrite in place) that also carries what the
    product's own loss heads can use to skip work: ``stacked`` — the one tensor the entries are unbind() views of;
    ``stream`` — the CUDA stream the entries were produced on (the sampling heads run on their own stream);
    ``neg_sim`` — on ``sim_score``: the similarity of every position with the batch-reversed language vector
    (train_DCNet.py:623-627), produced in the same pass as ``sim_score``.  Callers that ignore the attributes lose nothing."""
    __slots__ = ("stacked", "stream", "neg_sim")

    def __init__(self, items, stacked=None, stream=None, neg_sim=None):
        super().__init__(items)
        self.stacked, self.stream, self.neg_sim = stacked, stream, neg_sim


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

    def forward(self, x_nhwc, amax=None):
        """``amax`` / the ``_dcn_amax`` attribute of the input: abs-max word of x (ops.amax_*); the output carries its own."""
        amax = amax if amax is not None else getattr(x_nhwc, "_dcn_amax", None)
        if x_nhwc.dtype == torch.bfloat16 and not ops.storage_b16():
            x_nhwc = ops.to_f32(x_nhwc)          # (a bf16 producer in front of an fp32 block: mixed-precision experiments, ops.region)
        out, a = ConvBNAct.apply(x_nhwc, self.conv.weight, self.bn.weight, self.bn.bias, self.bn,
                                 self.conv.kernel_size[0], self.training, self.slope, amax, self.__dict__.get("_dcn_bank"),
                                 bool(self.__dict__.get("_dcn_out_b16")))          # (bf16-storage mode: the next block reads bf16)
        out._dcn_amax = a
        return out


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
        input_labels = input_labels.contiguous()
        lengths = ops.row_lengths(input_labels)                                             # :150 (ids != 0).sum(1)
        n, L = input_labels.shape
        emb = self.input_dropout(Embedding.apply(input_labels, self.embedding.weight))      # :168-169
        lin = self.mlp[0]
        embedded = LinearAct.apply(emb.view(n * L, -1), lin.weight, lin.bias, True).view(n, L, -1)     # :170
        r = self.rnn
        output = BiLSTM.apply(embedded, lengths,
                              r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0,
                              r.weight_ih_l0_reverse, r.weight_hh_l0_reverse, r.bias_ih_l0_reverse, r.bias_hh_l0_reverse)
        embedded = embedded * (torch.arange(L, device=embedded.device)[None, :]
                               < lengths[:, None]).unsqueeze(2).to(embedded.dtype)              # :178 pad_packed zeros
        # sentence vector = output[i, len_i - 1] (:185-188).  torch.gather: its backward is a scatter-add on unique rows — advanced
        # indexing (`output[arange, len - 1]`) would put a rocprim radix sort into the backward of every step
        sent = output.gather(1, (lengths - 1).clamp(min=0).view(n, 1, 1).expand(n, 1, output.size(2))).squeeze(1)
        return sent, output, embedded


class PhraseAttention(nn.Module):
    """model/DCNet_model.py:190-219 (csrc/phrase.hip).  grounding_model.forward evaluates its two instances in one launch
    together with the F.normalize that follows them; this standalone forward keeps the reference's module contract."""

    def __init__(self, input_dim):
        super().__init__()
        self.fc = nn.Linear(input_dim, 1)

    def forward(self, context, embedded, input_labels):
        weighted, _, attn = PhraseAttn.apply(context, embedded, input_labels, self.fc.weight, self.fc.bias, None, None, False)
        return attn[0], weighted


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
        if light:                                                # model/DCNet_model.py:296-312
            self.fcn_emb = nn.Sequential(OrderedDict([
                (str(i), nn.Sequential(ConvBatchNormReLU(embin_size, emb_size, 1, 1, 0, 1, leaky=leaky))) for i in range(3)]))
            self.fcn_out = nn.Sequential(OrderedDict([
                (str(i), nn.Sequential(nn.Conv2d(emb_size, 3 * 5, kernel_size=1))) for i in range(3)]))
        else:                                                    # :313-338
            self.fcn_emb = nn.Sequential(OrderedDict([
                (str(i), nn.Sequential(ConvBatchNormReLU(embin_size, emb_size, 1, 1, 0, 1, leaky=leaky),
                                       ConvBatchNormReLU(emb_size, emb_size, 3, 1, 1, 1, leaky=leaky),
                                       ConvBatchNormReLU(emb_size, emb_size, 1, 1, 0, 1, leaky=leaky))) for i in range(3)]))
            self.fcn_out = nn.Sequential(OrderedDict([
                (str(i), nn.Sequential(ConvBatchNormReLU(emb_size, emb_size // 2, 1, 1, 0, 1, leaky=leaky),
                                       nn.Conv2d(emb_size // 2, 3 * 5, kernel_size=1))) for i in range(3)]))
        self._coord_cache = {}
        self._pinned = {}
        self.sampler_busy_s = 0.0  # seconds the sampling worker threads spent drawing (accumulated)
        self._spec = None          # draws made ahead for the next training forward (_presample_take)
        # device tensors (sample_buffers) that already hold this forward's draws (draw_samples): the forward then neither
        # draws nor uploads — a captured training step reads its draws from these static buffers
        self.static_samples = None
        # "mt" (default): the negatives of both sampling heads come from Python's global MT19937 stream, bit-exact with the reference's
        # random.sample loops (csrc/sampling.cpp, host worker thread).  "device": drawn on the device by a counter-based generator
        # (csrc/sample.hip dcn_device_sample; SURVEY H3 option (ii)) — same distribution and exclusion rules, NOT the reference's numbers,
        # no host work per step (at 256 images the exact loop costs 0.19 s of a core per step: as long as the GPU step of configs[4]).
        # Python's `random` stream is not touched in this mode.
        self.sampler = "mt"
        self.sampler_seed = 0x5DC0E7A1         # device mode: seed of the generator (a data-parallel driver adds its rank)
        self._dev_sampler = {}
        self._streams = {}
        # the two sampling heads (K9, K14) on their own stream under the head convs of scales 1 and 2 (forward()), and
        # their contrastive losses on that stream too (losses.total_loss), so that their backward overlaps as well
        self.sampling_stream = True
        # a training forward also makes the draws of the NEXT one on its worker thread (used if Python's stream is still where
        # they started from: _presample_take); False = every forward draws for itself
        self.presample_ahead = True
        # the language branch (embedding, MLP, BiLSTM, mapping_lang, phrase attention) on its own stream under the backbone
        self.language_stream = True
        # a step driver that calls finish_backward() after loss.backward() sets this (graph.GraphedTrainStep): the language branch's
        # backward is then NOT part of loss.backward() — the head hands its gradients to detached leaves and finish_backward() runs the
        # branch's backward on its stream from the point where the last of them was written.  In a captured step that makes it the LAST
        # captured dependent of that point: the graph executor keeps it off the main chain's queue (tools/graph_sched.py) and its
        # latency-bound kernels (the persistent BiLSTM backward: 1.5 ms alone) run beside the backbone's backward, not in front of it
        self.defer_language_backward = False
        # discrete choices of the last forward (top-k / arg-max indices and the sampled negatives):
        # exposed so that parity tests can replay them through the oracle (near-tie orderings differ
        # between fp32 implementations) and so that a caller can log what was sampled
        self.last_choices = {}

    # ------------------------------------------------------------------------------------------
    def finish_backward(self) -> None:
        """Second half of a backward pass when ``defer_language_backward`` is set: the language branch's backward, on its own stream,
        from the gradients loss.backward() left on the detached leaves.  The caller's stream waits for it."""
        st = self.__dict__.pop("_lang_pending", None)
        if st is None:
            return
        side = st["side"]
        main = torch.cuda.current_stream()
        pairs = [(t_, l_.grad) for t_, l_ in zip(st["live"], st["leaves"]) if t_.requires_grad and l_.grad is not None]
        hops = []
        # one event per leaf, each recorded on the stream its AccumulateGrad ran on: `context` is also read by the sampling heads on
        # their stream, so its gradient is complete behind THAT stream's kernels, not behind the head's on the main one
        events = list(st["events"].values())
        st["event"] = events[-1] if events else None     # the last one written: the point the branch (and the hops) fork from
        if st["event"] is not None:
            # (captured steps: the graph executor gives the k-th dependent of a node the node's queue + k, of four — tools/graph_sched.py.
            #  The main chain is dependent 0, the weight-gradient stream sits on queue + 1: LANGUAGE_BWD_HOPS one-word memsets on streams of
            #  their own, waiting on the same event, take the places in between so that this branch gets a queue to itself)
            if torch.cuda.is_current_stream_capturing():
                for i_ in range(int(ops.LANGUAGE_BWD_HOPS_B16 if ops.storage_b16() else ops.LANGUAGE_BWD_HOPS)):
                    h_ = self._side_stream(main.device, "hop%d" % i_)
                    h_.wait_event(st["event"])
                    with torch.cuda.stream(h_):
                        self._hop_word(main.device).zero_()
                    hops.append(h_)
            for ev_ in reversed(events):
                side.wait_event(ev_)
        else:
            side.wait_stream(main)
        if pairs:
            with torch.cuda.stream(side):
                for _, g_ in pairs:
                    g_.record_stream(side)
                torch.autograd.backward([t_ for t_, _ in pairs], [g_ for _, g_ in pairs])
        main.wait_stream(side)
        for h_ in hops:
            main.wait_stream(h_)
        for l_ in st["leaves"]:
            l_.grad = None

    def _hop_word(self, device):
        key = ("hop_word", str(device))
        if key not in self._streams:
            self._streams[key] = torch.zeros(1, dtype=torch.int32, device=device)
        return self._streams[key]

    def _side_stream(self, device, name: str = "lang"):
        key = (str(device), name)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def _head_filter_banks(self):
        """The filter banks of the head's ConvBatchNormReLU blocks (mapping_visu, corr_conv, fcn_emb, fcn_out) in their GEMM forms,
        refreshed by ONE dcn_prepare_filters call per forward — as the backbone's (Darknet._filter_banks): per layer and step that
        replaces an OIHW->OHWI transpose, an abs-max pass and a pre-split in the forward and a filter transpose + pre-split in the
        backward (~150 launches of 5-10 us).  The first fcn_emb block (1032 input channels: the fusion layer) keeps its own path."""
        if not ops.FILTER_BANKS or not (ops.use_amax() or ops.storage_b16()):      # (bf16 storage reads the banks' bf16 forms)
            for blk in self._head_blocks():
                blk.__dict__["_dcn_bank"] = None
            return
        blocks = self._head_blocks()
        ws = {i: b.conv.weight for i, b in enumerate(blocks)}
        fb = self.__dict__.get("_hbanks")
        if fb is None or not fb.valid_for(ws):
            fb = ops.FilterBanks({i: w.detach() for i, w in ws.items()}, next(iter(ws.values())).device)
            self.__dict__["_hbanks"] = fb
        fb.refresh()
        for i, b in enumerate(blocks):
            b.__dict__["_dcn_bank"] = fb.get(i, b.conv.weight)

    def _head_blocks(self):
        out = []
        for seq in (self.mapping_visu, self.corr_conv, self.fcn_emb, self.fcn_out):
            for m in seq.modules():
                if isinstance(m, ConvBatchNormReLU):
                    out.append(m)
        return out

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
        flang = L2Norm.apply(z)                                                 # :485-487 F.normalize(dim=1), csrc/score.hip
        return word_id, flang, context, embedded

    def _fusion_head(self, s: int, corr, flang):
        """fcn_emb[s] + fcn_out[s] on one scale: corr (B,H,W,E) -> outbox logits (B,H,W,32 = 15 + padding)  (:491-506)."""
        h, w = corr.shape[1], corr.shape[2]
        ops.region("fusion")
        blk0 = self.fcn_emb[s][0]                                                # [corr | tile(flang) | coord] -> 1x1
        one = ops.amax_const(corr.device, 1.0) if ops.use_amax() else None        # corr is L2-normalised: |x| <= 1
        z, za = FusionConvBNAct.apply(corr.contiguous(), flang, self._coord(h, w, corr.device), blk0.conv.weight,
                                      blk0.bn.weight, blk0.bn.bias, blk0.bn, self.training, one)
        z._dcn_amax = za
        chain = list(self.fcn_emb[s])[1:] + list(self.fcn_out[s])[:-1]           # (none of them with light=True)
        if ops.storage_b16() and self.training and chain and (not ops.FILTER_BANKS or (self.emb_size // 2) % 32):
            raise RuntimeError("bf16 storage: the head chain fcn_emb -> fcn_out runs on bf16 tensors and needs the prepared filter banks "
                               f"(ops.FILTER_BANKS) and emb_size // 2 a multiple of 32 (emb_size = {self.emb_size})")
        n_emb = len(self.fcn_emb[s]) - 1
        for i_, blk in enumerate(chain):
            if i_ == n_emb:
                ops.region("out")
            blk.__dict__["_dcn_out_b16"] = True          # bf16-storage mode: the chain fcn_emb[1:] -> fcn_out[0] -> bbox head stays in bf16
            z = blk(z)
        last = self.fcn_out[s][-1]
        if z.dtype == torch.bfloat16 and not ops.storage_b16():
            z = ops.to_f32(z)
        return ConvBias.apply(z, last.weight, last.bias, getattr(z, "_dcn_amax", None))    # (B,H,W,32): channels 15..31 are zero padding

    def _scale_pairs(self, s: int, raw_s, flang, flang_attn):
        """Everything of scale s that depends only on its backbone tap (pair semantics): mapping + norm
        (:356-359), co-attention + corr_conv (:449-468), normalise + sim (:469,530-535), fusion head.
        Returns (fv, corr, sim, neg_sim|None, logits (B,H,W,32))."""
        ops.region("mapping")
        fv = L2Norm.apply(self.mapping_visu[s](raw_s, self.visumodel._tap_amax[s]))
        ops.region("corr")
        one = ops.amax_const(raw_s.device, 1.0) if ops.use_amax() else None       # unit-norm features and their convex combinations
        corr_raw = self.corr_conv[s][0](CoAttentionPairs.apply(fv, self.temperature), one)
        corr, sim, neg_sim = NormScore.apply(corr_raw, flang_attn, self.training)
        return fv, corr, sim, neg_sim, self._fusion_head(s, corr, flang)

    def _scale_nframe(self, s: int, raw_s, flang, flang_attn, B: int, n_frame: int):
        """Scale s of the inference model: centre frame vs every other frame, mean of the normalised
        correspondence features (model/test_DCNet_model.py:299-332), then the shared head."""
        one = ops.amax_const(raw_s.device, 1.0) if ops.use_amax() else None
        fv = L2Norm.apply(self.mapping_visu[s](raw_s, self.visumodel._tap_amax[s]))
        _, h, w, e = fv.shape
        clips = fv.view(B, n_frame, h * w, e)
        ctr, acc = n_frame // 2, None                                            # :303
        for idx in range(n_frame):                                               # :312-320
            if idx == ctr:
                continue
            cat = CoAttentionCenter.apply(clips, ctr, idx, self.temperature).view(B, h, w, 2 * e)
            z = self.corr_conv[s][0](cat, one)
            if torch.is_grad_enabled() and z.requires_grad:
                # the reference's train branch of this model (test_DCNet_model.py:480-483; its scripts never take it): differentiable
                # mean of the normalised features — L2Norm has a backward, the in-place accumulate of the inference path has none
                zn = L2Norm.apply(z)
                acc = zn * (1.0 / (n_frame - 1)) if acc is None else acc + zn * (1.0 / (n_frame - 1))
            else:
                acc = NormAccumulate.apply(z, acc, 1.0 / (n_frame - 1))          # :277-280, mean :324-332
        corr = acc
        sim = RowDot.apply(corr, flang_attn, False)                              # :386-391
        return fv, corr, sim, None, self._fusion_head(s, corr, flang)

    def _coord_rows(self, grids, device):
        """(P,8) coordinate rows of the three scales, coarsest first (model/DCNet_model.py:565-567); identical for every image."""
        key = ("rows", tuple(grids), str(device))
        if key not in self._coord_cache:
            self._coord_cache[key] = torch.cat([self._coord(g[0], g[1], device).reshape(-1, 8) for g in grids], dim=0).contiguous()
        return self._coord_cache[key]

    def _head(self, sim_score, logits, flang_loc):
        """The cross-scale tail (:545-621) as one autograd node of HIP kernels (functions.HeadTail): objectness x similarity,
        location module (rank-8 form), min-max, confidence modulation, NCHW outbox.
        Returns (outbox[3] NCHW, loc_score[3], only_obj[3])."""
        dev = flang_loc.device
        coord = self._coord_rows([(l.shape[1], l.shape[2]) for l in logits], dev)
        le, lt = self.loc_embedding, self.loc_text_embedding
        res = HeadTail.apply(*logits, *sim_score, flang_loc, coord, le[0].weight, le[0].bias, le[1].weight, le[1].bias,
                             lt[0].weight, lt[0].bias, lt[1].weight, lt[1].bias, le[1], lt[1], self.training)
        return list(res[0:3]), list(res[3:6]), list(res[6:9])

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
            # until the queued backbone kernels drain).  Two sets, used in turn: the draws of the NEXT forward are made
            # (speculatively, see forward()) while the upload of this one may still be in flight.
            mk = lambda: (torch.empty((n // 2, top_k, neg_n), dtype=torch.int64).pin_memory(),
                          torch.empty((n, hw, neg_c), dtype=torch.int64).pin_memory(),
                          torch.empty(hw + 1, dtype=torch.int32).pin_memory(),
                          torch.empty(n * hw * neg_c, dtype=torch.int32).pin_memory())
            self._pinned[key] = {"sets": [mk(), mk()], "events": [None, None], "turn": 0}
        pin = self._pinned[key]
        which = pin["turn"]; pin["turn"] ^= 1
        k9, k14, csr_off, csr_src = pin["sets"][which]
        if pin["events"][which] is not None:
            pin["events"][which].synchronize()   # the upload of this set (two forwards ago) has completed
            pin["events"][which] = None
        st, arr = _mt_state()
        L = lib()
        box = {"err": None, "shape": (n, top_k, hw, neg_n, neg_c), "pin": pin, "which": which, "g0": g0}

        secs = ctypes.c_double(0.0)

        def work():
            # ONE native call (K9 draws, K14 draws, the inverse table of the K14 negatives by position of the last image: the gather's
            # backward is then a deterministic segmented sum on the device), timed inside with the thread's CPU clock: between two
            # ctypes calls the worker needs the interpreter lock, and while the main thread runs Python that is up to a switch interval
            # (5 ms) each time — the wall-clock figure of round 5 (40 ms on the driver box against 10-16 ms of draws) was mostly that wait
            try:
                L.mt_sample_step(arr.ctypes.data, n, top_k, hw, neg_n, neg_c, k9.data_ptr(), k14.data_ptr(), csr_off.data_ptr(),
                                 csr_src.data_ptr(), ctypes.byref(secs))
            except BaseException as e:       # re-raised on the caller's thread by _presample_join
                box["err"] = e
            box["busy_s"] = secs.value

        th = threading.Thread(target=work, daemon=True)
        th.start()
        return th, st, arr, (k9, k14, csr_off, csr_src), box

    def _presample_take(self, n, g0):
        """The draws of this forward: the ones made ahead by the previous training forward if they are still valid — same
        shapes, and nobody has touched Python's global stream since (its state is the one they started from) — else fresh
        ones.  The K14 loop of the reference is O(N^2 HW) draws (11 M samples at 256 images: 0.4 s of one core, more than the
        backbone's forward), all of which must be made to leave the stream where the reference leaves it; made ahead, they run
        under the previous step's backward."""
        spec, self._spec = self._spec, None
        if spec is not None:
            th, st, arr, bufs, box = spec
            if box["shape"][0] == n and box["g0"] == g0 and random.getstate() == st:
                return spec
            th.join()                            # stale (re-seeded stream, other batch): let the worker finish, drop its draws
        return self._presample_start(n, g0)

    def device_samples(self, n: int, g0: int, device, top_k=30, neg_n=10, neg_c=5) -> dict:
        """``sampler = "device"``: this forward's draws, made by three kernels on the current stream (csrc/sample.hip
        dcn_device_sample) into buffers that live with the model — a captured step replays the kernels, and the generator's step
        counter lives on the device, so every replay draws new negatives.  Same tensors as ``_presample_join`` hands out."""
        hw = g0 * g0
        if n < 2 or hw - 1 < neg_n or hw - 1 < neg_c or hw * hw < top_k:
            raise ValueError(f"correspondence sampling needs >= 2 images and a coarsest grid of more than {max(neg_n, neg_c)} "
                             f"cells with >= {top_k} cell pairs (got {n} images, {g0}x{g0})")
        key = (torch.device(device).index, n, hw, top_k, neg_n, neg_c)
        st = self._dev_sampler.get(key)
        if st is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("grounding_model.device_samples: first use inside a stream capture (run one eager step first)")
            st = dict(self.sample_buffers(n, device, top_k, neg_n, neg_c, g0=g0))
            st["state"] = torch.tensor([int(self.sampler_seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
            st["ws"] = torch.zeros(int(lib().device_sample_ws(hw)), dtype=torch.int32, device=device)
            self._dev_sampler[key] = st
        lib().device_sample(st["state"].data_ptr(), n, top_k, hw, neg_n, neg_c, st["k9"].data_ptr(), st["k14"].data_ptr(),
                            st["csr_off"].data_ptr(), st["csr_src"].data_ptr(), st["ws"].data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
        return {k_: st[k_] for k_ in ("k9", "k14", "csr_off", "csr_src")}

    def sample_buffers(self, n: int, device, top_k=30, neg_n=10, neg_c=5, g0=None) -> dict:
        """Device tensors with the shapes of one training forward's draws (n images): the static buffers of a captured step."""
        hw = (self.img_size // 32 if g0 is None else g0) ** 2
        return {"k9": torch.zeros((n // 2, top_k, neg_n), dtype=torch.int64, device=device),
                "k14": torch.zeros((n, hw, neg_c), dtype=torch.int64, device=device),
                "csr_off": torch.zeros(hw + 1, dtype=torch.int32, device=device),
                "csr_src": torch.zeros(n * hw * neg_c, dtype=torch.int32, device=device)}

    def draw_samples(self, n: int, into: dict) -> dict:
        """Make the draws of ONE training forward on ``n`` images — Python's global MT19937 stream advances exactly as that
        forward would advance it — and upload them into the device tensors ``into`` (sample_buffers) on the current stream.
        With ``self.static_samples = into`` the forward then reads them instead of drawing itself: the host half of a step
        whose device half is a replayed hipGraph (dcnet_amd.graph.GraphedTrainStep)."""
        handle = self._presample_take(n, self.img_size // 32)
        return self._presample_join(handle, into["k9"].device, ahead=self.presample_ahead, into=into)

    def _presample_join(self, handle, device, ahead=False, into=None):
        th, st, arr, (k9, k14, csr_off, csr_src), box = handle
        th.join()
        if box["err"] is not None:
            raise box["err"]
        self.sampler_busy_s += box.get("busy_s", 0.0)      # what the native draws cost the worker thread (not the wait for the GPU)
        if random.getstate() != st:
            # someone drew from the global stream while the worker was running (another thread: forward itself draws
            # nothing else).  The worker started from a stale state: redo the draws from the current one, synchronously.
            st, arr = _mt_state()
            n, top_k, hw, neg_n, neg_c = box["shape"]
            L = lib()
            L.mt_sample_interframe(arr.ctypes.data, 0, n // 2, top_k, hw, neg_n, k9.data_ptr())
            L.mt_sample_crossmodal(arr.ctypes.data, n, hw, neg_c, k14.data_ptr())
            L.mt_sample_crossmodal_csr(k14.data_ptr(), n, hw, neg_c, csr_off.data_ptr(), csr_src.data_ptr())
        _mt_restore(st, arr)
        if device is None:                   # eval mode: the draws only advance the RNG stream, as in the reference
            return None
        if into is not None:
            for k_, src in (("k9", k9), ("k14", k14), ("csr_off", csr_off), ("csr_src", csr_src)):
                into[k_].copy_(src, non_blocking=True)
            out = into
        else:
            out = {"k9": k9.to(device, non_blocking=True), "k14": k14.to(device, non_blocking=True),
                   "csr_off": csr_off.to(device, non_blocking=True), "csr_src": csr_src.to(device, non_blocking=True)}
        ev = torch.cuda.Event(); ev.record()
        box["pin"]["events"][box["which"]] = ev
        if ahead:
            # the next training forward will most likely look the same: make its draws now, from the state the stream has
            # after this one's (checked again when they are taken)
            self._spec = self._presample_start(box["shape"][0], box["g0"])
        return out

    def _interframe_sampling(self, fv0, presampled, top_k=30):
        """model/DCNet_model.py:381-430 on the NHWC scale-0 map (N,g,g,E): csrc/sample.hip."""
        frame, corr, negf, index, neg_idx = InterframeSample.apply(fv0, presampled["k9"], top_k)
        self.last_choices["k9_index"] = index
        self.last_choices["k9_neg"] = neg_idx
        # the reference's lists are zero-copy unbinds of the stacked tensors
        return frame, corr, negf

    def _crossmodal(self, fv0, context, presampled):
        """model/DCNet_model.py:625-637 + Crossmodal_corrspondence :41-112: csrc/sample.hip."""
        fm = self.feature_map[0]
        vit, lag_pos, neg_cross, cols = CrossModalSample.apply(fv0, context, fm.weight, fm.bias, presampled["k14"],
                                                               presampled["csr_off"], presampled["csr_src"])
        self.last_choices["k14_cols"] = cols
        self.last_choices["k14_neg"] = presampled["k14"]
        return vit, lag_pos, neg_cross

    def _phrases(self, context, embedded, word_id):
        """sub_attn and loc_attn (:525,:556) with their F.normalize (:526,:557) in one launch."""
        sa, la = self.sub_attn.fc, self.loc_attn.fc
        flang_attn, flang_loc, _ = PhraseAttn.apply(context, embedded, word_id, sa.weight, sa.bias, la.weight, la.bias, True)
        return flang_attn, flang_loc

    # ------------------------------------------------------------------------------------------
    def forward(self, image, word_id, word_mask=None, n_frame: Optional[int] = None):
        if not image.is_cuda:
            raise RuntimeError("dcnet_amd.grounding_model runs on an MI355X only (HIP kernels, no CPU path)")
        if n_frame is not None:
            return self._forward_nframe(image, word_id, n_frame)
        if not self.training:
            return self._forward_pairs(image, word_id, word_mask)
        ops.batches_begin()                      # (the head's num_batches_tracked counters: one launch at the end instead of 23)
        try:
            return self._forward_pairs(image, word_id, word_mask)
        finally:
            ops.batches_end()

    def _forward_pairs(self, image, word_id, word_mask=None):
        N = image.size(0)
        if N % 2:
            raise ValueError("the training model consumes frame pairs: batch must be even (model/DCNet_model.py:365)")
        # The forward issues no host synchronisation at all (lengths are handled on the device), so the
        # host can queue step k+1 while the GPU still runs step k.  The language branch (latency-bound small
        # kernels) goes on a side stream: it runs under the backbone, and autograd replays its backward on the
        # same side stream under the backbone's backward.
        main = torch.cuda.current_stream()
        if ops.use_amax():
            ops.amax_begin_step(image.device)        # (before any stream forks: the step's abs-max words are zeroed on `main`)
        side = self._side_stream(image.device) if self.language_stream else main
        static = self.static_samples if self.training else None
        # In a captured step (only the dependencies count, not the order of the calls) the language branch is queued BEHIND the backbone's
        # first layers: its persistent BiLSTM holds 64 KB of LDS on half the CUs for 1.2 ms, and beside the one-workgroup-per-CU
        # kernels of those layers that costs them a second round (nconv1_kernel 0.62 -> 1.2 ms).  Eager: in front, so that the host's
        # 20 ms of backbone launches do not delay it.
        late = (ops.LANGUAGE_LATE and side is not main and self.training and torch.cuda.is_current_stream_capturing())

        def language():
            with torch.cuda.stream(side):
                w_, fl_, ctx_, emb_ = self._language(word_id)
                fa_, floc_ = self._phrases(ctx_, emb_, w_)    # :525-526, :556-557
                self._head_filter_banks()            # (0.3 ms of kernels the head needs after the backbone: beside it, not behind it)
            return w_, fl_, ctx_, emb_, fa_, floc_

        if not late:
            side.wait_stream(main)
            ops.region("language")
            word_id, flang, context, embedded, flang_attn, flang_loc = language()
        if self.sampler not in ("mt", "device"):
            raise ValueError(f"grounding_model.sampler = {self.sampler!r}: 'mt' or 'device'")
        on_device = self.sampler == "device"
        if on_device:
            static = None                        # (the draws are kernels of this forward: a captured step replays them, fresh every step)
        handle = None if (static is not None or on_device) else self._presample_take(N, image.shape[-1] // 32)   # worker thread, under the backbone (or made ahead)
        ops.region("backbone")
        raw = self.visumodel.forward_nhwc(image, taps_b16=True)                  # :344  (queued asynchronously; bf16-storage mode: bf16 taps)
        if late:
            ev = self.visumodel.__dict__.get("_early_event")
            if ev is not None:
                side.wait_event(ev)
            else:
                side.wait_stream(main)
            word_id, flang, context, embedded, flang_attn, flang_loc = language()
        main.wait_stream(side)
        for t_ in (flang, context, embedded, flang_attn, flang_loc):
            t_.record_stream(main)
        self.__dict__.pop("_lang_pending", None)
        if self.defer_language_backward and self.training and torch.is_grad_enabled() and side is not main:
            live = (flang, context, embedded, flang_attn, flang_loc)
            leaves = tuple(t_.detach().requires_grad_(t_.requires_grad) for t_ in live)
            state = {"live": live, "leaves": leaves, "events": {}, "side": side}

            def _written(leaf, st=state):        # (runs on the stream of the leaf's AccumulateGrad: the head's, or the sampling heads')
                ev_ = torch.cuda.Event(); ev_.record(torch.cuda.current_stream())
                st["events"].pop(id(leaf), None); st["events"][id(leaf)] = ev_       # insertion order = order of completion

            for l_ in leaves:
                if l_.requires_grad:
                    l_.register_post_accumulate_grad_hook(_written)
            self.__dict__["_lang_pending"] = state
            flang, context, embedded, flang_attn, flang_loc = leaves
        sampling = self.training
        samp = None
        r0 = self._scale_pairs(0, raw[0], flang, flang_attn)
        if sampling:
            # The two correspondence-sampling heads (:381-430, :625-637) read only the scale-0 features.  They go on their own
            # stream as soon as scale 0 is queued and run under the head convs of scales 1 and 2; autograd replays their
            # backward on the same stream, beside the heads' backward.  In eval mode the reference computes and discards
            # them: here only the RNG stream is advanced (the draws), the device work is skipped.
            if not on_device:
                presampled = static if static is not None else self._presample_join(handle, image.device, ahead=self.presample_ahead)
            samp = self._side_stream(image.device, "samp") if self.sampling_stream else main
            samp.wait_stream(main)
            with torch.cuda.stream(samp):
                if on_device:
                    presampled = self.device_samples(N, image.shape[-1] // 32, image.device)
                frame, corrf, negf = self._interframe_sampling(r0[0], presampled)
                vit, lag_pos, neg_cross = self._crossmodal(r0[0], context, presampled)
            for t_ in (r0[0], context, *presampled.values()):
                t_.record_stream(samp)
        elif not on_device:
            self._presample_join(handle, None)
        res = [r0, self._scale_pairs(1, raw[1], flang, flang_attn), self._scale_pairs(2, raw[2], flang, flang_attn)]
        corr_feat = [r[1] for r in res]
        sim = OutputList([r[2] for r in res], neg_sim=[r[3] for r in res] if sampling else None)
        ops.region("tail")
        outbox, loc, only_obj = self._head(sim, [r[4] for r in res], flang_loc)
        if not self.training:
            return outbox, sim, loc, only_obj
        if samp is not main:
            main.wait_stream(samp)               # callers read the sampled lists on the current stream
            for t_ in (frame, corrf, negf, vit, lag_pos, neg_cross):
                t_.record_stream(main)
        st = samp if samp is not main else None
        mk = lambda t: OutputList(t.unbind(1), stacked=t, stream=st)
        return (outbox, sim, loc, [c.permute(0, 3, 1, 2) for c in corr_feat], flang_attn.view(N, -1, 1, 1),
                mk(frame), mk(corrf), mk(negf), mk(vit), mk(lag_pos), mk(neg_cross))

    def _forward_nframe(self, image, word_id, n_frame: int):
        """model/test_DCNet_model.py:284-483."""
        if image.size(0) % n_frame:
            raise ValueError("batch must be a multiple of n_frame (model/test_DCNet_model.py:287)")
        B = image.size(0) // n_frame
        main = torch.cuda.current_stream()
        if ops.use_amax():
            ops.amax_begin_step(image.device)
        side = self._side_stream(image.device) if self.language_stream else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            word_id, flang, context, embedded = self._language(word_id)
            flang_attn, flang_loc = self._phrases(context, embedded, word_id)
        raw = self.visumodel.forward_nhwc(image, taps_b16=True)
        self._head_filter_banks()
        main.wait_stream(side)
        for t_ in (flang, context, embedded, flang_attn, flang_loc):
            t_.record_stream(main)
        res = [self._scale_nframe(s_, raw[s_], flang, flang_attn, B, n_frame) for s_ in range(3)]
        corr_feat = [r[1] for r in res]; sim = [r[2] for r in res]
        outbox, loc, only_obj = self._head(sim, [r[4] for r in res], flang_loc)
        corr_nchw = [c.permute(0, 3, 1, 2) for c in corr_feat]
        if self.training:
            return outbox, sim, loc, corr_nchw, flang_attn.view(B, -1, 1, 1)
        return outbox, sim, loc, corr_nchw, only_obj
