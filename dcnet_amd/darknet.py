"""Darknet-53 / YOLOv3 visual backbone on the HIP conv engine.

Mirrors the reference's model/darknet.py surface — ``parse_model_config``, ``create_modules``,
``Darknet(config_path, img_size, obj_out)`` with ``.module_list`` (same sub-module names, hence the
same 438 state_dict keys), ``.forward(x)`` returning the three taps, ``.load_weights`` /
``.save_weights`` (darknet binary) — but executes a static layer plan with hand-written kernels:

  * the graph is compiled once into a plan (dead YOLO heads removed: F7 of SURVEY.md; every
    shortcut fused into the producing conv; upsample fused with the route concat);
  * the whole backbone is ONE autograd node (``_DarknetFn``) whose backward is scheduled by hand:
    BN+LeakyReLU backward, weight gradient, data gradient per layer, gradients of tensors with two
    consumers accumulated inside the data-gradient kernel (no autograd add nodes);
  * eval mode folds BatchNorm into the conv epilogue: one kernel per layer.

Reference: model/darknet.py:99-116 (cfg parser), :162-237 (module factory), :391-431 (forward),
:433-513 (weights io); graph: model/yolov3.cfg.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ops

# ----------------------------------------------------------------------------------------------------
# cfg handling
# ----------------------------------------------------------------------------------------------------

def parse_model_config(path: str) -> List[dict]:
    """darknet .cfg -> list of block dicts (first block is [net]).  Same contract as the
    reference parser (model/darknet.py:99-116): '#' lines dropped, values kept as strings,
    ``batch_normalize`` defaults to 0 for (yolo)convolutional blocks."""
    blocks: List[dict] = []
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            if line.startswith("["):
                blocks.append({"type": line[1:-1].rstrip()})
                if blocks[-1]["type"] in ("convolutional", "yoloconvolutional"):
                    blocks[-1]["batch_normalize"] = 0
            else:
                key, value = line.split("=", 1)
                blocks[-1][key.strip()] = value.strip()
    return blocks


_ANCHORS = "10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326"


def yolov3_blocks(width: int = 416, height: int = 416) -> List[dict]:
    """The YOLOv3 graph DCNet uses, built from the network's regular structure (52-conv
    Darknet-53 trunk, then three neck/head groups with ``yoloconvolutional`` taps).  Produces
    the same block list as parsing the reference's model/yolov3.cfg."""
    blocks: List[dict] = [dict(type="net", batch="1", subdivisions="1", width=str(width), height=str(height),
                               channels="3", momentum="0.9", decay="0.0005")]

    def conv(filters, size, stride=1, bn=1, act="leaky", kind="convolutional"):
        b = {"type": kind, "batch_normalize": bn} if bn else {"type": kind, "batch_normalize": 0}
        if bn:
            b["batch_normalize"] = "1"
        b.update(filters=str(filters), size=str(size), stride=str(stride), pad="1", activation=act)
        blocks.append(b)

    def shortcut():
        blocks.append({"type": "shortcut", "from": "-3", "activation": "linear"})

    conv(32, 3)
    for filters, reps in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        conv(filters, 3, 2)
        for _ in range(reps):
            conv(filters // 2, 1); conv(filters, 3); shortcut()
    for mask, (filters, lateral) in zip(("6,7,8", "3,4,5", "0,1,2"), ((512, None), (256, 61), (128, 36))):
        if lateral is not None:
            blocks.append({"type": "route", "layers": "-4"})
            conv(filters, 1)
            blocks.append({"type": "upsample", "stride": "2"})
            blocks.append({"type": "route", "layers": f"-1, {lateral}"})
        for _ in range(2):
            conv(filters, 1); conv(filters * 2, 3)
        conv(filters, 1, kind="yoloconvolutional")
        conv(filters * 2, 3)
        conv(255, 1, bn=0, act="linear")
        blocks.append({"type": "yolo", "mask": mask, "anchors": _ANCHORS, "classes": "80", "num": "9",
                       "jitter": ".3", "ignore_thresh": ".7", "truth_thresh": "1", "random": "1"})
    return blocks


def write_cfg(path: str, blocks: Optional[Sequence[dict]] = None) -> None:
    blocks = yolov3_blocks() if blocks is None else blocks
    with open(path, "w") as fh:
        for b in blocks:
            fh.write(f"[{b['type']}]\n")
            for k, v in b.items():
                if k != "type" and not (k == "batch_normalize" and str(v) == "0"):
                    fh.write(f"{k}={v}\n")
            fh.write("\n")


class EmptyLayer(nn.Module):
    """Placeholder for route / shortcut slots (model/darknet.py:239-243)."""


class MyUpsample2(nn.Module):
    """Nearest x2 (model/darknet.py:158-160); only a slot marker here — the plan fuses it."""
    def forward(self, x):
        return x[:, :, :, None, :, None].expand(-1, -1, -1, 2, -1, 2).reshape(x.size(0), x.size(1), x.size(2) * 2, x.size(3) * 2)


class YOLOLayer(nn.Module):
    """Detection layer slot (model/darknet.py:245-375).  DCNet discards its output
    (obj_out=False), so it is never executed; it has no parameters."""
    def __init__(self, anchors, num_classes, img_dim):
        super().__init__()
        self.anchors, self.num_classes, self.image_dim = anchors, num_classes, img_dim


def create_modules(module_defs: List[dict]):
    """Parameter containers with the reference's names (model/darknet.py:162-237):
    ``conv_%d`` (bias only when there is no BN), ``batch_norm_%d``, ``leaky_%d``."""
    hyper = module_defs.pop(0)
    filters_out = [int(hyper["channels"])]
    module_list = nn.ModuleList()
    for i, d in enumerate(module_defs):
        seq = nn.Sequential()
        t = d["type"]
        if t in ("convolutional", "yoloconvolutional"):
            bn = int(d["batch_normalize"]); filters = int(d["filters"]); k = int(d["size"])
            pad = (k - 1) // 2 if int(d["pad"]) else 0
            seq.add_module(f"conv_{i}", nn.Conv2d(filters_out[-1], filters, k, int(d["stride"]), pad, bias=not bn))
            if bn:
                seq.add_module(f"batch_norm_{i}", nn.BatchNorm2d(filters))
            if d["activation"] == "leaky":
                seq.add_module(f"leaky_{i}", nn.LeakyReLU(0.1))
        elif t == "upsample":
            assert int(d["stride"]) == 2
            seq.add_module(f"upsample_{i}", MyUpsample2())
            filters = filters_out[-1]
        elif t == "route":
            layers = [int(x) for x in d["layers"].split(",")]
            filters = sum(filters_out[1:][l] if l >= 0 else filters_out[l] for l in layers)
            seq.add_module(f"route_{i}", EmptyLayer())
        elif t == "shortcut":
            filters = filters_out[int(d["from"])]
            seq.add_module(f"shortcut_{i}", EmptyLayer())
        elif t == "yolo":
            idx = [int(x) for x in d["mask"].split(",")]
            a = [int(x) for x in d["anchors"].split(",")]
            a = [(a[j], a[j + 1]) for j in range(0, len(a), 2)]
            seq.add_module(f"yolo_{i}", YOLOLayer([a[j] for j in idx], int(d["classes"]), 256))
            filters = filters_out[-1]
        else:
            raise ValueError(f"unsupported cfg block type {t!r}")
        module_list.append(seq)
        filters_out.append(filters)
    return hyper, module_list


# ----------------------------------------------------------------------------------------------------
# plan
# ----------------------------------------------------------------------------------------------------

class _ConvOp:
    __slots__ = ("slot", "src", "dst", "res", "cin", "cout", "k", "stride", "bn", "leaky", "need_dx")


class _UpCatOp:
    __slots__ = ("dst", "up_src", "lat_src", "c_up", "c_lat")


class _AliasOp:
    __slots__ = ("dst", "src")


def build_plan(module_defs: List[dict], in_channels: int = 3):
    """Compile the slot list into executable ops.  Returns (ops, tap_slots, channels_per_slot)."""
    n = len(module_defs)
    ch: List[int] = []
    prev = in_channels
    for i, d in enumerate(module_defs):
        t = d["type"]
        if t in ("convolutional", "yoloconvolutional"):
            prev = int(d["filters"])
        elif t == "route":
            ls = [int(x) for x in d["layers"].split(",")]
            prev = sum(ch[l if l >= 0 else i + l] for l in ls)
        elif t == "shortcut":
            prev = ch[i + int(d["from"])]
        ch.append(prev)
    # liveness: what can reach a tap (input of each yoloconvolutional)
    taps = [i - 1 for i, d in enumerate(module_defs) if d["type"] == "yoloconvolutional"]
    live = [False] * n
    stack = list(taps)
    while stack:
        i = stack.pop()
        if i < 0 or live[i]:
            continue
        live[i] = True
        d = module_defs[i]; t = d["type"]
        if t in ("convolutional", "yoloconvolutional", "upsample"):
            stack.append(i - 1)
        elif t == "shortcut":
            stack += [i - 1, i + int(d["from"])]
        elif t == "route":
            stack += [l if l >= 0 else i + l for l in (int(x) for x in d["layers"].split(","))]
    consumers: Dict[int, List[int]] = {i: [] for i in range(-1, n)}
    for i, d in enumerate(module_defs):
        if not live[i]:
            continue
        t = d["type"]
        if t in ("convolutional", "yoloconvolutional", "upsample"):
            consumers[i - 1].append(i)
        elif t == "shortcut":
            consumers[i - 1].append(i); consumers[i + int(d["from"])].append(i)
        elif t == "route":
            for l in (int(x) for x in d["layers"].split(",")):
                consumers[l if l >= 0 else i + l].append(i)
    plan = []
    fused_shortcuts = set(); fused_upsamples = set()
    for i, d in enumerate(module_defs):
        if not live[i]:
            continue
        t = d["type"]
        if t in ("convolutional", "yoloconvolutional"):
            op = _ConvOp()
            op.slot = i; op.src = i - 1; op.dst = i; op.res = None
            op.cin = in_channels if i == 0 else ch[i - 1]; op.cout = int(d["filters"])
            op.k = int(d["size"]); op.stride = int(d["stride"])
            op.bn = bool(int(d["batch_normalize"])); op.leaky = d["activation"] == "leaky"
            op.need_dx = i > 0
            nxt = module_defs[i + 1] if i + 1 < n else None
            if (nxt is not None and nxt["type"] == "shortcut" and live[i + 1] and consumers[i] == [i + 1]
                    and i not in taps):
                op.dst = i + 1; op.res = i + 1 + int(nxt["from"])
                fused_shortcuts.add(i + 1)
            plan.append(op)
        elif t == "shortcut":
            if i not in fused_shortcuts:
                raise NotImplementedError("unfused shortcut (pattern not present in YOLOv3)")
        elif t == "upsample":
            nxt = module_defs[i + 1]
            ok = nxt["type"] == "route" and consumers[i] == [i + 1]
            ls = [int(x) for x in nxt["layers"].split(",")] if ok else []
            if not (ok and len(ls) == 2 and ls[0] == -1):
                raise NotImplementedError("upsample not followed by route(-1, k)")
            fused_upsamples.add(i + 1)
            op = _UpCatOp()
            op.dst = i + 1; op.up_src = i - 1; op.lat_src = ls[1] if ls[1] >= 0 else i + 1 + ls[1]
            op.c_up = ch[i - 1]; op.c_lat = ch[op.lat_src]
            plan.append(op)
        elif t == "route":
            if i in fused_upsamples:
                continue
            ls = [int(x) for x in d["layers"].split(",")]
            if len(ls) != 1:
                raise NotImplementedError("route concat without upsample")
            op = _AliasOp(); op.dst = i; op.src = ls[0] if ls[0] >= 0 else i + ls[0]
            plan.append(op)
    return plan, taps, ch


# ----------------------------------------------------------------------------------------------------
# execution
# ----------------------------------------------------------------------------------------------------

def _cpad(c: int) -> int:
    return 4 if c <= 4 else ops.pad32(c)


def _run_forward(plan, taps, x_nhwc, P, training: bool, save: Optional[dict], banks=None, taps_b16: bool = False):
    """P[slot] = dict(w=OIHW weight, b=conv bias|None, gamma, beta, rm, rv).  Returns tap tensors.
    banks (ops.FilterBanks, refreshed by the caller): the filter banks of all slots in their GEMM forms."""
    out: Dict[int, torch.Tensor] = {-1: x_nhwc}
    # abs-max word of every activation (ops.amax_*): written by the kernel that produces the tensor, read by the GEMMs
    # that consume it (forward here, weight gradient in the backward) to pick their power-of-two operand scales
    am = ops.use_amax()
    # bf16 storage (ops.set_precision("bf16s")): every activation / raw conv output behind the stem is a bf16 tensor; the three taps are
    # cast to fp32 for the head (and their gradients back to bf16 in _run_backward)
    s16 = ops.storage_b16()
    amx: Dict[int, Optional[torch.Tensor]] = {-1: None}
    # outputs with exactly ONE reader, a train-mode conv + BatchNorm whose kernels can apply the activation while loading
    # (ops.pre_supported): they are handed on as ops.PreAct — no scale_act pass, no activation tensor
    readers: Dict[int, list] = {}
    f8_left: Dict[int, int] = {}          # fp8 storage: e4m3 readers of a tensor still to come (its attached copy is dropped after the last)
    for op in plan:
        ins = (op.src, op.res) if isinstance(op, _ConvOp) else (op.up_src, op.lat_src) if isinstance(op, _UpCatOp) else (op.src,)
        for s_ in ins:
            if s_ is not None:
                readers.setdefault(s_, []).append(op)

    def sole_pre_reader(op, x):
        """the only reader of op's output, when that is a train-mode conv + BatchNorm whose kernels take the raw tensor"""
        rd = readers.get(op.dst, [])
        if not (am and ops.PRE_ACT and training and save is not None and len(rd) == 1 and op.dst not in taps and op.res is None):
            return False
        nx = rd[0]
        ho, wo = ops.conv_out_hw(x.shape[1], x.shape[2], op.k, op.stride)
        return (isinstance(nx, _ConvOp) and nx.bn and nx.src == op.dst and nx.res != op.dst
                and ops.pre_supported(x.shape[0], ho, wo, nx.cin, nx.cout, nx.k, nx.stride))

    # the last layer of the persistent one-workgroup-per-CU kernels (csrc/nconv.hip: the 32 <-> 64 channel 3x3 layers): an event behind
    # it lets other streams start their LDS-heavy work (the persistent BiLSTM) where it no longer takes CUs from those launches
    early = [i for i, op in enumerate(plan) if isinstance(op, _ConvOp) and op.k == 3 and op.cin == 32 and op.cout == 64]
    early_at = early[-1] if early else -1
    early_event = None
    for i_op, op in enumerate(plan):
        if i_op == early_at + 1 and early_at >= 0:
            early_event = torch.cuda.Event(); early_event.record()
        if isinstance(op, _ConvOp):
            p = P[op.slot]
            x = out[op.src]; ax = amx.get(op.src)
            bank = banks.get(op.slot, p["w"]) if banks is not None else None
            if bank is not None and getattr(banks, "pending", None) is not None:
                torch.cuda.current_stream().wait_stream(banks.pending)      # the refresh launched by Darknet._filter_banks
                banks.pending = None
            if bank is not None:       # prepared for the whole network in one go (no per-layer transpose / abs-max / pre-split)
                w, aw, wsp, w16 = bank["ohwi"], (bank["amax"] if am else None), bank["split"], bank["b16"]
                w._dcn_wt = (bank["t"], bank["tsplit"]); w._dcn_wt16 = bank["tb16"]; w._dcn_bank = bank
            else:
                w = ops.weight_to_ohwi(p["w"]); wsp = w16 = None
                aw = ops.absmax(p["w"]) if (am and op.cin > 4) else None
            ao = ops.amax_slot(x.device) if am else None
            res = out[op.res] if op.res is not None else None
            act = ops.ACT_LEAKY if op.leaky else ops.ACT_NONE
            if s16 and ops.B16_DIAG is not None and not training and op.cin > 4:
                # experiments (tools/precision_criterion.py --backbone): which part of the bf16-storage backbone costs the box criterion.
                # "layers": predicate on the conv slot — True = this layer on bf16 tensors, False = on fp32 tensors at fp32 accuracy;
                # "res32": the shortcut sums are kept in fp32 (the convolutions read a bf16 copy)
                diag = ops.B16_DIAG
                on16 = diag.get("layers", lambda s_: True)(op.slot)
                if not on16:
                    ops.set_precision(diag.get("arith", "fp32"))      # ("bf16": fp32 TENSORS, operands rounded to bf16 in the loaders)
                    try:
                        ss = ops.bn_fold(p["gamma"], p["beta"], p["rm"], p["rv"], 1e-5) if op.bn else (None, p["b"])
                        # (no abs-max words: the launch takes the bf16 three-piece split — fp32 accuracy all the same)
                        o, _ = ops.conv2d_fwd(ops.to_f32(x), w, op.k, op.stride, ss[0], ss[1], act, 0.1,
                                              residual=None if res is None else ops.to_f32(diag.get("_r32", {}).get(op.res, res)))
                    finally:
                        ops.set_precision("bf16s")
                    if res is not None and diag.get("res32"):
                        diag.setdefault("_r32", {})[op.dst] = o
                    out[op.dst] = o; amx[op.dst] = None
                    continue
                x = ops.to_b16(x)
                if res is not None and diag.get("res32"):
                    ss = ops.bn_fold(p["gamma"], p["beta"], p["rm"], p["rv"], 1e-5)
                    o32, _ = ops.conv2d_fwd_b16(x, bank["b16"], op.cout, op.k, op.stride, ss[0], ss[1], act, 0.1, out_f32=True)
                    r32 = diag.setdefault("_r32", {})
                    o32 = o32 + (r32[op.res] if op.res in r32 else ops.to_f32(res))
                    r32[op.dst] = o32
                    out[op.dst] = ops.to_b16(o32); amx[op.dst] = None
                    continue
                res = None if res is None else ops.to_b16(res)
            if s16 and x.dtype == torch.bfloat16:
                if bank is None:
                    raise RuntimeError("bf16 storage needs the prepared filter banks (ops.FILTER_BANKS) for every layer behind the stem")
                use8 = ops.f8_takes(op.cin, op.cout, op.k)          # "fp8s": this layer's forward on e4m3 operands (quantised here, once)
                if use8:
                    x8, xs = ops.quant_of(x)                         # (the copy scale_act wrote beside x, else a pass now)
                    # the copy rides on the tensor, which the backward keeps (its weight gradient reads the bf16 values): drop it once
                    # the last e4m3 reader has it (round-5 advice: ~0.3 GB of dead bytes at 64 images otherwise)
                    f8_left[op.src] = f8_left.get(op.src, sum(1 for r_ in readers.get(op.src, []) if isinstance(r_, _ConvOp)
                                                              and ops.f8_takes(r_.cin, r_.cout, r_.k))) - 1
                    if f8_left[op.src] <= 0 and hasattr(x, "_dcn_q8"):
                        del x._dcn_q8
                    w8, ws = ops.bank_q8(bank, "q8", bank["b16"], op.cout)       # (made once per step by FilterBanks.refresh)
                if op.bn and training:
                    if use8:
                        y, stats = ops.conv2d_fwd_f8(x8, xs, w8.view(-1), ws, op.cout, op.k, op.stride, want_stats=True)
                    else:
                        y, stats = ops.conv2d_fwd_b16(x, bank["b16"], op.cout, op.k, op.stride, want_stats=True)
                    mi = ops.bn_finalize(stats, y.numel() // op.cout, p["gamma"], p["beta"], 1e-5, p["momentum"], p["rm"], p["rv"])
                    # "fp8s": a reader that takes e4m3 operands gets its copy from this pass (no quantisation pass of its own)
                    q_ = ops.storage_f8() and any(isinstance(r_, _ConvOp) and r_.src == op.dst and ops.f8_takes(r_.cin, r_.cout, r_.k)
                                                  for r_ in readers.get(op.dst, []))
                    o = ops.scale_act(y, mi[2], mi[3], act, 0.1, residual=res, quant=q_)
                    if save is not None:
                        save[op.slot] = (x, y, mi, w, None, None)
                else:
                    if op.bn:
                        ss = ops.bn_fold(p["gamma"], p["beta"], p["rm"], p["rv"], 1e-5)
                        scale, shift = ss[0], ss[1]
                    else:
                        scale, shift = None, p["b"]
                    if use8:
                        o, _ = ops.conv2d_fwd_f8(x8, xs, w8.view(-1), ws, op.cout, op.k, op.stride, scale, shift, act, 0.1, residual=res)
                    else:
                        o, _ = ops.conv2d_fwd_b16(x, bank["b16"], op.cout, op.k, op.stride, scale, shift, act, 0.1, residual=res)
                out[op.dst] = o; amx[op.dst] = None
                continue
            if op.bn and training:
                # (the conv hands out the abs-max of its raw output with its store: the bound below starts from it)
                ay = ops.amax_slot(x.device) if (not isinstance(x, ops.PreAct) and sole_pre_reader(op, x)) else None
                y, stats = ops.conv2d_fwd(x, w, op.k, op.stride, want_stats=True, amax_x=ax, amax_w=aw, amax_out=ay,
                                          w_split_ready=wsp, w_b16=w16)
                cnt = y.numel() // op.cout
                mi = ops.bn_finalize(stats, cnt, p["gamma"], p["beta"], 1e-5, p["momentum"], p["rm"], p["rv"])
                if ay is not None and y.is_contiguous():
                    ao = ops.bn_act_amax_bound(ay, mi[2], mi[3], 0.1)
                    o = ops.PreAct(y, mi[2], mi[3], act, 0.1)
                    if ops.PRE_ACT == "check":        # (tests: the activation written out, read with the same abs-max word)
                        o = o.materialise()
                else:
                    o = ops.scale_act(y, mi[2], mi[3], act, 0.1, residual=res, amax_out=ao, out_b16=s16)    # (s16: the stem's fp32 raw output -> bf16)
                if save is not None:
                    save[op.slot] = (x, y, mi, w, ax, aw)
            else:
                if op.bn:
                    ss = ops.bn_fold(p["gamma"], p["beta"], p["rm"], p["rv"], 1e-5)
                    scale, shift = ss[0], ss[1]
                else:
                    scale, shift = None, p["b"]
                if save is None:      # inference: one kernel per layer, shortcut fused in the epilogue
                    o, _ = ops.conv2d_fwd(x, w, op.k, op.stride, scale, shift, act, 0.1, residual=res, amax_x=ax, amax_w=aw, amax_out=ao,
                                          w_split_ready=wsp, w_b16=w16)
                else:                 # frozen-BN fine-tuning: keep the pre-shortcut activation for act'
                    if res is None:
                        a, _ = ops.conv2d_fwd(x, w, op.k, op.stride, scale, shift, act, 0.1, amax_x=ax, amax_w=aw, amax_out=ao, w_split_ready=wsp, w_b16=w16)
                        o = a
                    else:
                        a, _ = ops.conv2d_fwd(x, w, op.k, op.stride, scale, shift, act, 0.1, amax_x=ax, amax_w=aw, w_split_ready=wsp, w_b16=w16)
                        o = ops.scale_act(a, None, None, ops.ACT_NONE, 0.0, residual=res, amax_out=ao)
                    save[op.slot] = (x, a, scale, w, ax, aw)
            if s16:
                o = ops.to_b16(o)          # (the stem in inference: its fused epilogue wrote fp32)
            out[op.dst] = o; amx[op.dst] = ao
        elif isinstance(op, _UpCatOp):
            up, lat = out[op.up_src], out[op.lat_src]
            if up.dtype != lat.dtype:              # (mixed-precision experiments only)
                up = ops.to_b16(up) if lat.dtype == torch.bfloat16 else ops.to_f32(up)
            n, h, w_, _ = lat.shape
            buf = torch.empty((n, h, w_, op.c_up + op.c_lat), dtype=lat.dtype, device=lat.device)
            ops.upsample2_into(up, buf[..., :op.c_up])
            ops.copy_slice(lat, buf[..., op.c_up:])
            out[op.dst] = buf
            # the concat holds exactly the values of its two sources: its abs-max is the larger of theirs
            amx[op.dst] = ops.absmax(lat, ops.absmax(up)) if am else None
        else:
            out[op.dst] = out[op.src]; amx[op.dst] = amx.get(op.src)
    # (taps_b16: the caller's first kernels read bf16 — grounding_model's mapping convolutions — so the taps stay as they are)
    return [ops.to_f32(out[t]) if (s16 and not taps_b16) else out[t] for t in taps], [amx.get(t) for t in taps], early_event


def _run_backward(plan, taps, grads_taps, P, save, training: bool, sink=None, bucket_bytes: int = 0):
    """Hand-scheduled reverse sweep.  Returns {slot: dict(w=, b=, gamma=, beta=)} parameter grads.
    sink (optional): called with [(slot, key, tensor), ...] every time ``bucket_bytes`` of parameter gradients are final —
    the data-parallel reducer starts their all-reduce while the sweep continues (parallel.OverlappedGradReducer)."""
    pending: list = []
    pending_bytes = 0
    g: Dict[int, Optional[torch.Tensor]] = {}
    s16 = ops.storage_b16()

    def add(slot, t):
        cur = g.get(slot)
        if cur is None:
            if not t.is_contiguous():
                buf = torch.empty(t.shape, dtype=t.dtype, device=t.device)
                ops.copy_slice(t, buf)
                t = buf
            g[slot] = t
        else:
            ops.copy_slice(t, cur, accumulate=True)

    for t, gt in zip(taps, grads_taps):
        if gt is not None:
            g0 = gt.clone(memory_format=torch.contiguous_format)      # never accumulate into autograd's tensor
            add(t, ops.to_b16(g0) if s16 else g0)
    pg: Dict[int, dict] = {}
    # BatchNorm taps: the data gradient that completes the gradient of a conv + BN + activation output (= the FIRST consumer of that
    # output in plan order, processed last) can form the partial sums that layer's BatchNorm backward starts with (ops.conv2d_bwd_data)
    producer = {op.dst: op for op in plan if isinstance(op, _ConvOp)}
    first_use: Dict[int, int] = {}
    for i, op in enumerate(plan):
        ins = (op.src, op.res) if isinstance(op, _ConvOp) else (op.up_src, op.lat_src) if isinstance(op, _UpCatOp) else (op.src,)
        for s_ in ins:
            if s_ is not None:
                first_use.setdefault(s_, i)
    index_of = {id(op): i for i, op in enumerate(plan)}
    tapped: Dict[int, torch.Tensor] = {}
    held = None          # (gradient dict, slot, ops.HeldWgrad): the weight gradient of the layer before, launched behind this layer's passes

    def release():
        nonlocal held, pending_bytes
        ops.release_held_wgrads()        # (ops.WGRAD_DIRECT: the head's last block)
        if held is not None:
            d_, slot_, hw_ = held
            held = None
            d_["w"] = hw_.issue()
            if sink is not None:
                pending.append((slot_, "w", d_["w"])); pending_bytes += d_["w"].numel() * 4

    for op in reversed(plan):
        dout = g.pop(op.dst, None)
        if dout is None:
            continue
        if isinstance(op, _AliasOp):
            add(op.src, dout)
        elif isinstance(op, _UpCatOp):
            n, h2, w2, _ = dout.shape
            cur = g.get(op.up_src)
            if cur is None:
                cur = torch.empty((n, h2 // 2, w2 // 2, op.c_up), dtype=dout.dtype, device=dout.device)
                ops.upsample2_bwd(dout[..., :op.c_up], cur, False)
                g[op.up_src] = cur
            else:
                ops.upsample2_bwd(dout[..., :op.c_up], cur, True)
            add(op.lat_src, dout[..., op.c_up:])
        else:
            p = P[op.slot]
            x, y, aux, w, ax, aw = save.pop(op.slot)
            stem_fused = (op.bn and training and not op.need_dx and ops.STEM_FUSED_BWD and torch.is_tensor(x) and x.shape[3] == 4 and op.cout == 32
                          and op.k == 3 and op.stride == 1 and op.res is None and torch.is_tensor(y) and y.is_contiguous() and x.shape[2] >= 32)
            if s16 and torch.is_tensor(x) and x.dtype == torch.float32 and dout.dtype == torch.bfloat16 and not stem_fused:
                dout = ops.to_f32(dout)           # bf16 storage: the 3-channel stem's generic kernels stay fp32 (its fused backward reads bf16 itself)
            shape = tuple(p["w"].shape)
            d = {}
            ady = ops.amax_slot(dout.device) if ops.use_amax() else None
            if (op.bn and training and not op.need_dx and ops.STEM_FUSED_BWD and x.shape[3] == 4 and op.cout == 32 and op.k == 3
                    and op.stride == 1 and op.res is None and y.is_contiguous() and x.shape[2] >= 32):
                # the stem: nothing but its weight gradient reads dy, so dy is formed inside that kernel and never written
                mi = aux
                dw, d["gamma"], d["beta"] = ops.stem_bwd_weight_bn(x, y, dout, mi[0], mi[1], p["gamma"], p["beta"],
                                                                   ops.ACT_LEAKY if op.leaky else ops.ACT_NONE, 0.1,
                                                                   part=tapped.pop(op.slot, None))
                d["w"] = ops.weight_grad_to_oihw(dw, shape)
                pg[op.slot] = d
                if sink is not None:
                    for k_, t_ in d.items():
                        if t_ is not None:
                            pending.append((op.slot, k_, t_)); pending_bytes += t_.numel() * 4
                    if pending_bytes >= bucket_bytes:
                        sink(pending); pending = []; pending_bytes = 0
                continue
            if op.bn and training:
                mi = aux
                dy, dgamma, dbeta = ops.bn_act_bwd(y, dout, mi[0], mi[1], p["gamma"], p["beta"],
                                                   ops.ACT_LEAKY if op.leaky else ops.ACT_NONE, 0.1, amax_out=ady,
                                                   part=tapped.pop(op.slot, None),
                                                   quant=bool(op.need_dx) and ops.f8_takes(op.cout, op.cin, op.k))      # ("fp8s": dy's e4m3 copy for the data gradient)
                d["gamma"], d["beta"] = dgamma, dbeta
            else:
                # frozen statistics: y holds act(scale*conv+shift) before the shortcut add
                dz = ops.act_bwd(y, dout, 0.1) if op.leaky else dout
                if aux is not None:       # folded BN: z = scale*conv + shift
                    xhat_like = None
                    dy = dz * aux
                    d["beta"] = dz.reshape(-1, op.cout).sum(0)
                    # dgamma = sum(dz * (conv - rm) * rsqrt(rv+eps)); recover conv from y only where act is
                    # invertible (leaky/none are), z = y>0 ? y : y/slope
                    z = torch.where(y > 0, y, y / 0.1) if op.leaky else y
                    gsafe = torch.where(p["gamma"] == 0, torch.ones_like(p["gamma"]), p["gamma"])
                    d["gamma"] = (dz * (z - p["beta"]) / gsafe).reshape(-1, op.cout).sum(0)
                else:
                    dy = dz
                    if p["b"] is not None:
                        d["b"] = dz.reshape(-1, op.cout).sum(0)
                ady = None            # (frozen-BN path: the GEMMs below compute the abs-max of dy themselves)
            if op.res is not None:
                add(op.res, dout)
            release()                  # (behind this layer's BatchNorm passes: the main chain stays the first dependent of the data gradient before)
            if not ops.WGRAD_AFTER_DGRAD:
                d["w"] = ops.wgrad_on_side(x, dy, op.k, op.stride, shape, amax_x=ax, amax_dy=ady)     # overlaps with the data gradient below
            if op.need_dx and s16 and dy.dtype == torch.bfloat16:
                cur = g.get(op.src)
                tap = None
                prev = producer.get(op.src)
                if (ops.BN_TAP and ops.BN_TAP_TRUNK and training and op.stride == 1 and prev is not None and prev.bn
                        and first_use.get(op.src) == index_of[id(op)] and prev.slot in save):
                    _, y_prev, mi_prev = save[prev.slot][:3]
                    if torch.is_tensor(y_prev) and y_prev.dtype == torch.bfloat16 and y_prev.is_contiguous():
                        tap = dict(y=y_prev, mean=mi_prev[0], invstd=mi_prev[1], gamma=P[prev.slot]["gamma"], beta=P[prev.slot]["beta"],
                                   act=ops.ACT_LEAKY if prev.leaky else ops.ACT_NONE, slope=0.1)
                if ops.f8_takes(op.cout, op.cin, op.k) and dy.is_contiguous():      # "fp8s": the data gradient on e4m3 operands
                    dy8, dys = ops.quant_of(dy)
                    wt8, wts = ops.bank_q8(getattr(w, "_dcn_bank", None), "tq8", getattr(w, "_dcn_wt16"), op.cin)
                    res_ = ops.conv2d_bwd_data_f8(dy8, dys, wt8.view(-1), wts, (x.shape[1], x.shape[2]), x.shape[3], op.k, op.stride,
                                                  out=cur, accumulate=cur is not None, tap=tap)
                else:
                    res_ = ops.conv2d_bwd_data_b16(dy, getattr(w, "_dcn_wt16"), (x.shape[1], x.shape[2]), x.shape[3], op.k, op.stride,
                                                   out=cur, accumulate=cur is not None, tap=tap)
                if tap is not None:
                    res_, part_ = res_
                    if part_ is not None:
                        tapped[prev.slot] = part_
                if cur is None:
                    g[op.src] = res_
            elif op.need_dx:
                cur = g.get(op.src)
                hw = (x.shape[1], x.shape[2])
                wtr = getattr(w, "_dcn_wt", None)      # the transposed banks of this step (ops.FilterBanks)
                wt16 = getattr(w, "_dcn_wt16", None)
                tap = None
                prev = producer.get(op.src)
                # (the register-bank kernel of the stride-2 layer behind the stem; every stride-1 layer on conv1.hip / conv3.hip)
                if (ops.BN_TAP and training and prev is not None and prev.bn and first_use.get(op.src) == index_of[id(op)]
                        and prev.slot in save
                        and ((x.shape[3] == 32 and op.stride == 2 and op.k == 3) or (ops.BN_TAP_TRUNK and op.stride == 1 and x.shape[3] >= 64
                                 and (ops.BN_TAP_TRUNK is True or ops.BN_TAP_TRUNK == 1 or (ops.BN_TAP_TRUNK == 2) == (op.k == 3))))):
                    _, y_prev, mi_prev = save[prev.slot][:3]
                    if torch.is_tensor(y_prev) and y_prev.is_contiguous():
                        tap = dict(y=y_prev, mean=mi_prev[0], invstd=mi_prev[1], gamma=P[prev.slot]["gamma"], beta=P[prev.slot]["beta"],
                                   act=ops.ACT_LEAKY if prev.leaky else ops.ACT_NONE, slope=0.1)
                res_ = ops.conv2d_bwd_data(dy, w, hw, op.k, op.stride, out=cur, accumulate=cur is not None, amax_dy=ady, amax_w=aw,
                                           wt_ready=wtr, wt_b16=wt16, tap=tap)
                if tap is not None:
                    res_, part_ = res_
                    if part_ is not None:
                        tapped[prev.slot] = part_
                if cur is None:
                    g[op.src] = res_
            if ops.WGRAD_AFTER_DGRAD:      # (schedule experiment: queued behind the data gradient, beside the next layer's BatchNorm passes)
                if ops.WGRAD_SIDE and ops.WGRAD_HELD:
                    held = (d, op.slot, ops.HeldWgrad(x, dy, op.k, op.stride, shape, amax_x=ax, amax_dy=ady))
                else:
                    d["w"] = ops.wgrad_on_side(x, dy, op.k, op.stride, shape, amax_x=ax, amax_dy=ady)
            pg[op.slot] = d
            if sink is not None:
                for k_, t_ in d.items():
                    if t_ is not None:
                        pending.append((op.slot, k_, t_)); pending_bytes += t_.numel() * 4
                if pending_bytes >= bucket_bytes:
                    sink(pending); pending = []; pending_bytes = 0
    release()
    if sink is not None and pending:
        sink(pending)
    if pg:
        ops.join_side(next(iter(P.values()))["w"].device)
    return pg


class _DarknetFn(torch.autograd.Function):
    """The whole backbone as one autograd node.  Flat parameter order per conv slot:
    weight, (bias | gamma, beta)."""

    @staticmethod
    def forward(ctx, net: "Darknet", training: bool, image: torch.Tensor, *flat):
        plan, taps = net._plan, net._taps
        P = net._param_table(flat)
        x = ops.nchw_to_nhwc(image.contiguous(), 4)
        # (needs_input_grad speaks about the parameters, not about the caller's grad mode: under torch.no_grad() — inference — nothing is
        #  saved and every layer is ONE kernel with BatchNorm, activation and shortcut in its epilogue; forward_nhwc notes the mode, since
        #  inside a Function's forward grad mode is always off)
        need_grad = any(ctx.needs_input_grad[3:]) and bool(net.__dict__.get("_grad_on", True))
        ctx.no_backward_reason = None if need_grad else "nothing was saved: the forward ran with gradients off (or no parameter requires one)"
        if ops.storage_b16() and not training:
            need_grad = False          # bf16 storage has no frozen-BatchNorm backward: inference only in eval mode (backward() says so)
            ctx.no_backward_reason = ("bf16 storage has no backward in eval mode (frozen-BatchNorm fine-tuning is not built); "
                                      "train with .train(), or use the fp32 precision mode")
        save = {} if need_grad else None
        outs, tap_amax, net._early_event = _run_forward(plan, taps, x, P, training, save, net._filter_banks(P),
                                                        taps_b16=bool(net.__dict__.get("_taps_b16")))
        net._tap_amax = tap_amax              # abs-max words of the three taps (read by the head's first convolutions)
        if save is not None:
            # outputs must go through save_for_backward (an attribute reference would make a
            # ctx <-> output cycle and pin the whole activation set until the GC runs)
            for slot, tup in list(save.items()):
                for k, o in enumerate(outs):
                    if tup[0] is o:
                        save[slot] = (("tap", k),) + tup[1:]
            ctx.save_for_backward(*outs)
        ctx.net, ctx.training, ctx.save, ctx.nflat = net, training, save, len(flat)
        ctx.P = P
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        net = ctx.net
        if ctx.save is None:
            why = getattr(ctx, "no_backward_reason", None) or ("this node's saved tensors were consumed by an earlier backward "
                                                               "(one backward per forward; retain_graph is not supported)")
            raise (NotImplementedError if "bf16" in why else RuntimeError)("dcnet_amd.Darknet: " + why)
        outs = ctx.saved_tensors
        save = {slot: ((outs[t[0][1]],) + t[1:] if isinstance(t[0], tuple) else t) for slot, t in ctx.save.items()}
        ctx.save = None
        red = net.__dict__.get("_grad_reducer")       # parallel.OverlappedGradReducer (data-parallel runs), else None
        sink = None
        if red is not None:
            pmap = net._slot_params()
            sink = lambda items: red.push([(pmap[(slot, key)], t) for slot, key, t in items if (slot, key) in pmap])
        pg = _run_backward(net._plan, net._taps, grads, ctx.P, save, ctx.training, sink, red.bucket_bytes if red is not None else 0)
        if red is not None:
            red.join_backward()                        # the averaged gradients are complete before autograd sees them
        flat_grads: List[Optional[torch.Tensor]] = []
        for op in net._conv_ops:
            d = pg.get(op.slot, {})
            flat_grads.append(d.get("w"))
            if op.bn:
                flat_grads += [d.get("gamma"), d.get("beta")]
            else:
                flat_grads.append(d.get("b"))
        return (None, None, None) + tuple(flat_grads)


class Darknet(nn.Module):
    """YOLOv3 backbone (drop-in for model/darknet.py:377-431).  ``forward`` returns the list of
    taps [1024@S/32, 512@S/16, 256@S/8] in NCHW like the reference; ``forward_nhwc`` hands the
    NHWC tensors to the rest of this package without a layout round trip."""

    def __init__(self, config_path: str = "./model/yolov3.cfg", img_size: int = 416, obj_out: bool = False):
        super().__init__()
        if obj_out:
            raise NotImplementedError("obj_out=True (YOLO detection output) is outside the DCNet hot path")
        self.config_path = config_path
        self.obj_out = obj_out
        if config_path and os.path.exists(config_path):
            self.module_defs = parse_model_config(config_path)
        else:
            self.module_defs = yolov3_blocks()
        self.hyperparams, self.module_list = create_modules(self.module_defs)
        self.img_size = img_size
        self.seen = 0
        self.header_info = np.array([0, 0, 0, self.seen, 0], dtype=np.int32)
        self._plan, self._taps, self._ch = build_plan(self.module_defs, int(self.hyperparams["channels"]))
        self._conv_ops = [op for op in self._plan if isinstance(op, _ConvOp)]

    # -- parameter plumbing -------------------------------------------------------------------
    def _flat_params(self) -> List[torch.Tensor]:
        flat = []
        for op in self._conv_ops:
            seq = self.module_list[op.slot]
            conv = seq[0]
            flat.append(conv.weight)
            if op.bn:
                flat += [seq[1].weight, seq[1].bias]
            else:
                flat.append(conv.bias)
        return flat

    def _param_table(self, flat):
        P = {}
        it = iter(flat)
        for op in self._conv_ops:
            seq = self.module_list[op.slot]
            d = dict(w=next(it).detach(), b=None, gamma=None, beta=None, rm=None, rv=None, momentum=0.1)
            if op.bn:
                bn = seq[1]
                d["gamma"], d["beta"] = next(it).detach(), next(it).detach()
                d["rm"], d["rv"], d["momentum"] = bn.running_mean, bn.running_var, bn.momentum
            else:
                d["b"] = next(it).detach()
            P[op.slot] = d
        return P

    def _slot_params(self):
        """{(slot, "w" | "gamma" | "beta" | "b"): parameter} of the live convolutions."""
        out = {}
        for op in self._conv_ops:
            seq = self.module_list[op.slot]
            out[(op.slot, "w")] = seq[0].weight
            if op.bn:
                out[(op.slot, "gamma")] = seq[1].weight; out[(op.slot, "beta")] = seq[1].bias
            elif seq[0].bias is not None:
                out[(op.slot, "b")] = seq[0].bias
        return out

    def _filter_banks(self, P):
        """All filter banks of the backbone in their GEMM forms, refreshed once per forward (ops.FilterBanks: three launches
        instead of ~5 small kernels per layer; the banks of a step serve its backward too).  The job table holds raw parameter
        addresses, so it is rebuilt when a parameter moved."""
        if not ops.FILTER_BANKS:
            return None
        ws = {slot: d["w"] for slot, d in P.items()}
        fb = self.__dict__.get("_fbanks")
        if fb is None or not fb.valid_for(ws):
            fb = ops.FilterBanks(ws, next(iter(ws.values())).device)
            self.__dict__["_fbanks"] = fb
        # the refresh (0.6 ms of kernels) runs on a stream of its own, beside the image transpose and the stem, which do not read
        # the banks; _run_forward joins it in front of the first convolution that does
        main = torch.cuda.current_stream()
        prep = self.__dict__.get("_prep_stream")
        if prep is None or prep.device != main.device:
            prep = torch.cuda.Stream(device=main.device)
            self.__dict__["_prep_stream"] = prep
        prep.wait_stream(main)                        # (the parameters' last writer and the banks' last readers are on `main`)
        with torch.cuda.stream(prep):
            fb.refresh()
        fb.pending = prep
        return fb

    def forward_nhwc(self, x: torch.Tensor, taps_b16: bool = False) -> List[torch.Tensor]:
        """taps_b16 (bf16-storage mode only): hand the three taps out as the bf16 tensors they are instead of casting them to fp32."""
        self.__dict__["_taps_b16"] = bool(taps_b16) and ops.storage_b16()
        self.__dict__["_grad_on"] = torch.is_grad_enabled()
        if not x.is_cuda:
            raise RuntimeError("dcnet_amd.Darknet runs on an MI355X only: move the model and inputs to cuda "
                               "(there is no CPU path; the CPU restatement lives in oracle/ for tests)")
        training = self.training
        if training:
            torch._foreach_add_([self.module_list[op.slot][1].num_batches_tracked for op in self._conv_ops if op.bn], 1)
        return list(_DarknetFn.apply(self, training, x, *self._flat_params()))

    def forward(self, x, targets=None):
        if targets is not None:
            raise NotImplementedError("YOLO detection training (targets=...) is outside the DCNet hot path")
        return [t.permute(0, 3, 1, 2) for t in self.forward_nhwc(x)]

    # -- darknet binary weights (model/darknet.py:433-513) -----------------------------------
    def load_weights(self, weights_path: str) -> None:
        """[5 x int32 header] then per conv slot: (bn.bias, bn.weight, running_mean, running_var | conv.bias),
        conv.weight — for EVERY conv slot incl. the dead YOLO heads (file compatibility)."""
        with open(weights_path, "rb") as fp:
            header = np.fromfile(fp, dtype=np.int32, count=5)
            weights = np.fromfile(fp, dtype=np.float32)
        self.header_info = header
        self.seen = int(header[3])
        ptr = 0

        def take(t: torch.Tensor):
            nonlocal ptr
            n = t.numel()
            if ptr + n > weights.size:
                raise ValueError(f"{weights_path}: truncated weights file")
            t.data.copy_(torch.from_numpy(weights[ptr:ptr + n]).view_as(t))
            ptr += n

        for d, seq in zip(self.module_defs, self.module_list):
            if d["type"] in ("convolutional", "yoloconvolutional"):
                conv = seq[0]
                if int(d["batch_normalize"]):
                    bn = seq[1]
                    take(bn.bias); take(bn.weight); take(bn.running_mean); take(bn.running_var)
                else:
                    take(conv.bias)
                take(conv.weight)

    def save_weights(self, path: str, cutoff: int = -1, reference_layout: bool = False) -> None:
        """Writes every conv slot, i.e. the format ``load_weights`` (here and in the reference, model/darknet.py:433-483)
        reads.  The reference's own writer (model/darknet.py:490-513) skips the three ``yoloconvolutional`` slots, so its
        files cannot be read back by its loader; ``reference_layout=True`` reproduces that file byte for byte
        (pinned by oracle/make_format_goldens.py)."""
        hdr = np.array(self.header_info, dtype=np.int32).copy()
        hdr[3] = self.seen
        defs = self.module_defs if cutoff == -1 else self.module_defs[:cutoff]
        if reference_layout and cutoff == -1:
            defs = self.module_defs[:-1]                     # the reference slices [:cutoff] with cutoff = -1 (:498)
        kinds = ("convolutional",) if reference_layout else ("convolutional", "yoloconvolutional")
        with open(path, "wb") as fp:
            hdr.tofile(fp)
            for d, seq in zip(defs, self.module_list):
                if d["type"] in kinds:
                    conv = seq[0]
                    if int(d["batch_normalize"]):
                        bn = seq[1]
                        for t in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                            t.detach().cpu().numpy().astype(np.float32).tofile(fp)
                    else:
                        conv.bias.detach().cpu().numpy().astype(np.float32).tofile(fp)
                    conv.weight.detach().cpu().numpy().astype(np.float32).tofile(fp)
