"""Thin tensor-level wrappers over the C ABI (dcnet_amd.lib): torch supplies device memory and
the stream, every computation below is a call into libdcnet_hip.so.

Layout convention inside the package: activations NHWC (contiguous torch tensors of shape
(N,H,W,C)), conv weights OHWI (Cout,k,k,Cin_padded).  Nothing here has a fallback path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch

from .lib import DcnError, lib

ACT_NONE, ACT_LEAKY = 0, 1

# Matrix-pipe precision of the wide GEMM tiles (conv forward / data gradient / weight gradient, co-attention GEMMs):
#   "fp32"      fp32 accuracy on the f16 matrix pipe — the default: operands scaled by a per-tensor power of two (from the
#               tensor's tracked abs-max), cut into two f16 pieces (11 + 11 bits), three MFMAs per product; launches whose
#               operands carry no abs-max run as "fp32_bf16x3"
#   "fp32_bf16x3" fp32 accuracy on the bf16 matrix pipe (3 exact bf16 pieces per operand, 6 cross terms)
#   "fp32_mfma" the native fp32 MFMA instruction on every tile
#   "bf16"      bf16 operands (round to nearest even), fp32 accumulate: BASELINE.json configs[2]; reduced precision,
#               builder-defined (the reference has no bf16 semantics); tensors in HBM stay fp32
#   "fp8"       forward and data-gradient tiles with OCP fp8 e4m3 operands (per-tensor power-of-two scales from an abs-max
#               pass, fp32 accumulate), weight gradient with bf16 operands: BASELINE.json configs[4]; reduced precision
#   "bf16s"     bf16 STORAGE (configs[2] proper): the convolution stacks keep activations, raw conv outputs and their
#               gradients as bf16 tensors in HBM (fp32 accumulators, BatchNorm statistics, weight gradients and master
#               weights); their kernels are the *_b16 entry points (csrc/conv1.hip conv1b, wgrad.hip IN16, b16.hip) — one
#               MFMA per product, half the bytes per element.  What stays on fp32 tensors (the 3-channel stem, co-attention,
#               scoring, heads' tails, language branch) runs as in "bf16" (bf16 operands)
#   "fp8s"      fp8 STORAGE for the operands of the wide 3x3 convolutions (configs[4] proper, round 5) on top of "bf16s": their forward and
#               data gradient read OCP e4m3 tensors with one e8m0 scale per pixel / per filter (written by dcn_quant_rows_e4m3 from the bf16
#               tensor) and multiply on the block-scaled MFMA (2 x the bf16 rate, half the staged bytes); everything else as in "bf16s"
#               (bf16 tensors, bf16 weight gradients, fp32 accumulators / statistics / master weights)
PRECISIONS = {"fp32_mfma": 0, "fp32_bf16x3": 1, "bf16": 2, "fp8": 3, "fp32": 4, "bf16s": 2, "fp8s": 2}
_precision = "fp32"


def set_precision(mode: str) -> None:
    global _precision
    if mode not in PRECISIONS:
        raise ValueError(f"precision {mode!r}: expected one of {sorted(PRECISIONS)}")
    lib().set_tuning(b"precision", PRECISIONS[mode])
    _precision = mode


def storage_b16() -> bool:
    """True in the bf16-storage modes ("bf16s", and "fp8s" which adds fp8 operand tensors): conv stacks allocate and exchange bf16 tensors."""
    return _precision in ("bf16s", "fp8s")


F8_MIN_K = 128           # channels on the contraction side from which a 3x3 convolution takes fp8 operands (64: also the 104-wide layers)


def storage_f8() -> bool:
    return _precision == "fp8s"


def f8_takes(k_channels: int, out_channels: int, ksize: int) -> bool:
    """Does a convolution pass with `k_channels` on its contraction side (forward: Cin, data gradient: Cout) run on fp8 operands in the
    "fp8s" mode?  The multi-tap layers only: their launches gather every row nine times and are bound by the bytes they stage — where 1-byte
    operands pay for the quantisation pass in front (a 1x1 layer reads its input once: the pass would cost what it saves)."""
    # (measured in one process, replayed steps, ms: 54.47 as is; the backbone's 1x1 layers as well, their operand copies written by the
    #  producing passes: 54.45 — nothing: they are bound by HBM bytes that do not change; the 64-channel 3x3 layers as well: 55.04)
    return storage_f8() and ksize == 3 and k_channels % 64 == 0 and k_channels >= F8_MIN_K and out_channels % 32 == 0


def _b16(t) -> bool:
    return t is not None and t.dtype == torch.bfloat16


def _chk16(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous bf16 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


def _rows16(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype in (torch.bfloat16, torch.float32) and _rows_ok(t) and t.stride(-2) % 8 == 0 and t.shape[-1] % 8 == 0
            and t.data_ptr() % 16 == 0):
        raise ValueError(f"{name}: expected a bf16 / fp32 CUDA [rows][c] view, c and row stride multiples of 8, 16-byte aligned; got "
                         f"{t.dtype} shape {tuple(t.shape)} strides {t.stride()}")


def cast_rows(src: torch.Tensor, dst: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """dst[..., :c] (+)= src[..., :c] between fp32 / bf16 [rows][c] views (csrc/b16.hip): casts at the fp32 boundary of the
    bf16-storage mode, channel-slice copies of the route concat."""
    c = src.shape[-1]
    rows = src.numel() // c
    _rows16(src, "cast_rows src"); _rows16(dst, "cast_rows dst")
    if dst.shape[-1] != c or dst.numel() // c != rows:
        raise ValueError("cast_rows: shapes differ")
    lib().cast_rows(src.data_ptr(), int(_b16(src)), src.stride(-2), dst.data_ptr(), int(_b16(dst)), dst.stride(-2), rows, c,
                    int(accumulate), _s())
    return dst


def to_b16(x: torch.Tensor) -> torch.Tensor:
    return x if _b16(x) else cast_rows(x, torch.empty(x.shape, dtype=torch.bfloat16, device=x.device))


def to_f32(x: torch.Tensor) -> torch.Tensor:
    return cast_rows(x, torch.empty(x.shape, dtype=torch.float32, device=x.device)) if _b16(x) else x


def get_precision() -> str:
    return _precision


# Experiments only (tools/precision_criterion.py --regions): a precision mode per region of the model's forward, to localise where a
# reduced-precision mode loses the box criterion.  None (the product's state) = one mode for everything, region() is a no-op.
REGION_PRECISION = None


B16_DIAG = None      # experiments: {"layers": slot -> bool, "res32": bool} for the bf16-storage backbone in eval mode (darknet._run_forward)


def region(name: str) -> None:
    """Called by grounding_model at the start of a region (language, backbone, mapping, corr, fusion, out, tail)."""
    if REGION_PRECISION is not None:
        set_precision(REGION_PRECISION.get(name, REGION_PRECISION.get("default", "fp32")))


# ---- abs-max words of GEMM operands (the f16 two-piece split derives its power-of-two scales from them) --------------
# A word holds the float bits of max|tensor| and is written with order-independent atomic maxima by the kernels that
# produce the tensor (scale_act, bn_act_bwd, the conv epilogue) or by absmax().  Words come zeroed from a pool (one per
# forward, amax_begin_step) and are used once per step, so a saved activation keeps its word for the backward.
AMAX_WORDS = 64          # DCN_AMAX_WORDS: the waves of a producer spread their atomic maxima over this many words
_amax_pools = {}
_amax_recent = {}
_amax_consts = {}


def use_amax() -> bool:
    """The precision modes whose GEMM tiles scale their operands by a power of two derived from the tensors' abs-max words:
    the f16 two-piece split ("fp32") and the fp8 path (whose separate abs-max passes, dcn_f8_scale, they replace)."""
    return _precision in ("fp32", "fp8")      # ("bf16s": no operand scaling at all)


AMAX_SLOTS = 2048        # slots of one pool = what a forward (and its backward) may hand out before a second pool is taken


def _amax_new_pool(device, sync: bool):
    """[pool tensor, cursor (slots), end (slots)], zero-filled on the current stream.  ``sync``: the caller is not on a stream
    every later user forks from (op-level calls outside a step) — the device is synchronised once so that no stream can see the
    memset land after its first atomic maximum."""
    t = torch.zeros(AMAX_SLOTS * AMAX_WORDS, dtype=torch.int32, device=device)
    if sync and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize(device)
    return [t, 0, AMAX_SLOTS]


def amax_begin_step(device) -> None:
    """Call at the start of a forward, on the stream every other stream of the step forks from: the step gets a pool of its
    OWN (512 KB from torch's caching allocator, zeroed on that stream), so the words it hands out are zero before any of its
    kernels (on whichever stream) updates them.  A saved activation holds a view of its word and thereby the pool: however many
    forwards — training micro-batches, validation passes — run before a backward, its words are never recycled under it (the
    round-3 form rotated four quarters of one pool and zeroed the oldest).  Under hipGraph capture the allocation and the memset
    are part of the graph: every replay starts from zeroed words."""
    key = torch.device(device).index
    pool = _amax_new_pool(device, sync=False)
    _amax_pools[key] = pool
    if not torch.cuda.is_current_stream_capturing():
        # side streams read a step's words too: the block goes back to the allocator no earlier than four forwards later, long
        # after every stream of its step has joined the main one
        recent = _amax_recent.setdefault(key, [])
        recent.append(pool[0])
        del recent[:-4]


def amax_slot(device) -> torch.Tensor:
    key = torch.device(device).index
    pool = _amax_pools.get(key)
    if pool is None or pool[1] >= pool[2]:
        # no step has begun (op-level tests) or the pool is used up (> 2048 GEMM operands in one step): take another one — the
        # old one stays alive through the slots that reference it
        pool = _amax_new_pool(device, sync=True)
        _amax_pools[key] = pool
    t = pool[0][pool[1] * AMAX_WORDS:(pool[1] + 1) * AMAX_WORDS]
    pool[1] += 1
    return t


def amax_const(device, value: float) -> torch.Tensor:
    """A word holding a KNOWN bound (e.g. 1.0 for L2-normalised features): no pass over the data."""
    key = (torch.device(device).index, float(value))
    t = _amax_consts.get(key)
    if t is None:
        t = torch.full((AMAX_WORDS,), value, dtype=torch.float32).view(torch.int32).to(device)
        _amax_consts[key] = t
    return t


def absmax(x: torch.Tensor, slot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """slot = max(slot, max|x|) for an fp32 [rows][c] view (or any contiguous tensor); returns the slot."""
    if slot is None:
        slot = amax_slot(x.device)
    if x.is_contiguous() and x.numel() % 4 == 0:
        rows, c, ld = 1, x.numel(), x.numel()
        if c > (1 << 30):
            c = x.shape[-1]; rows = x.numel() // c; ld = c
    else:
        _rows(x, "absmax")
        c = x.shape[-1]; rows = x.numel() // c; ld = x.stride(-2)
    lib().absmax(x.data_ptr(), rows, c, ld, slot.data_ptr(), _s())
    return slot


def _amax_or_pass(t: torch.Tensor, given: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if not use_amax():
        return None
    return given if given is not None else absmax(t)


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous fp32 CUDA tensor, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")


def _rows_ok(t: torch.Tensor) -> bool:
    """True if t, read as [rows][c] with row stride t.stride(-2), is addressed uniformly."""
    if t.stride(-1) != 1:
        return False
    for i in range(t.dim() - 2):
        if t.shape[i] != 1 and t.stride(i) != t.stride(i + 1) * t.shape[i + 1]:
            return False
    return True


def _rows(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and _rows_ok(t)):
        raise ValueError(f"{name}: expected an fp32 CUDA [rows][c] view with one uniform row stride, got shape "
                         f"{tuple(t.shape)} strides {t.stride()}")


# ---- scratch -----------------------------------------------------------------------------------
_scratch = {}


def scratch(n_floats: int, device, slot: int = 0) -> torch.Tensor:
    """A per-device, per-slot grow-only fp32 scratch buffer (ops are stream-ordered, so one
    buffer per slot can be shared by every call on the stream)."""
    # one buffer per (device, slot, stream): kernels on different streams may run concurrently
    key = (torch.device(device).index, slot, torch.cuda.current_stream().cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < n_floats:
        buf = torch.empty(max(int(n_floats), 1 << 20), dtype=torch.float32, device=device)
        _scratch[key] = buf
    return buf


_slab_counters = {}
SLAB_COUNTERS = 4096      # include/dcnet_hip.h DCN_SLAB_COUNTERS


def slab_counters(device, slot: int = 0) -> torch.Tensor:
    """The split-K arrival counters of the weight-gradient launches of one (device, slot, stream): zero at allocation, left zero by
    every launch (dcn_conv2d_bwd_weight).  Keyed like ``scratch``: launches that share a slab workspace share the counters, and are
    ordered on one stream."""
    key = (torch.device(device).index, slot, torch.cuda.current_stream().cuda_stream)
    buf = _slab_counters.get(key)
    if buf is None:
        buf = torch.zeros(SLAB_COUNTERS, dtype=torch.int32, device=device)
        _slab_counters[key] = buf
    return buf


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


# ---- layout ------------------------------------------------------------------------------------
def nchw_to_nhwc(x: torch.Tensor, c_pad: Optional[int] = None) -> torch.Tensor:
    _chk(x, "nchw_to_nhwc")
    n, c, h, w = x.shape
    c_pad = c if c_pad is None else c_pad
    out = torch.empty((n, h, w, c_pad), dtype=torch.float32, device=x.device)
    lib().nchw_to_nhwc(x.data_ptr(), out.data_ptr(), n, c, h, w, c_pad, _s())
    return out


def nhwc_to_nchw(x: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    _chk(x, "nhwc_to_nchw")
    n, h, w, ld = x.shape
    c = ld if c is None else c
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    lib().nhwc_to_nchw(x.data_ptr(), out.data_ptr(), n, c, h, w, ld, _s())
    return out


def weight_to_ohwi(w: torch.Tensor, ci_pad: Optional[int] = None, co_pad: Optional[int] = None) -> torch.Tensor:
    """OIHW parameter -> OHWI kernel operand.  The 3-channel stem becomes the packed [Co][64]
    form (9 taps x 4 channels, zero padded) that the c4 path of the conv engine reads."""
    w = w.detach()
    _chk(w, "weight_to_ohwi")
    co, ci, kh, kw = w.shape
    if ci <= 4:
        tmp = torch.empty((co, kh, kw, 4), dtype=torch.float32, device=w.device)
        lib().oihw_to_ohwi(w.data_ptr(), tmp.data_ptr(), co, ci, kh, kw, 4, _s())
        out = torch.zeros((co, 64), dtype=torch.float32, device=w.device)
        out[:, :kh * kw * 4] = tmp.view(co, -1)
        return out
    ci_pad = pad32(ci) if ci_pad is None else ci_pad
    co_pad = co if co_pad is None else co_pad
    if kh == 1 and kw == 1 and ci_pad == ci and co_pad == co:
        return w.view(co, 1, 1, ci)            # a 1x1 filter bank is the same bytes in OIHW and OHWI: no kernel
    alloc = torch.zeros if co_pad != co else torch.empty
    out = alloc((co_pad, kh, kw, ci_pad), dtype=torch.float32, device=w.device)
    lib().oihw_to_ohwi(w.data_ptr(), out.data_ptr(), co, ci, kh, kw, ci_pad, _s())
    return out


def weight_grad_to_oihw(dw: torch.Tensor, shape: Tuple[int, int, int, int]) -> torch.Tensor:
    """Adjoint of weight_to_ohwi: OHWI gradient (possibly channel/filter padded) -> OIHW."""
    co, ci, kh, kw = shape
    if ci <= 4:
        tmp = dw[:, :kh * kw * 4].contiguous().view(co, kh, kw, 4)
        out = torch.empty(shape, dtype=torch.float32, device=dw.device)
        lib().ohwi_to_oihw(tmp.data_ptr(), out.data_ptr(), co, ci, kh, kw, 4, _s())
        return out
    _chk(dw, "weight_grad_to_oihw")
    ci_pad = dw.shape[3]
    if kh == 1 and kw == 1 and ci_pad == ci and dw.shape[0] == co:
        return dw.view(co, ci, 1, 1)
    out = torch.empty(shape, dtype=torch.float32, device=dw.device)
    lib().ohwi_to_oihw(dw.data_ptr(), out.data_ptr(), co, ci, kh, kw, ci_pad, _s())
    return out


# ---- convolution ---------------------------------------------------------------------------------
def conv_out_hw(h: int, w: int, k: int, stride: int) -> Tuple[int, int]:
    pad = (k - 1) // 2
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def f8_scales(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Device tensor {s_a, s_w}: power-of-two scales that map max|a| and max|w| into the fp8 e4m3 range (dcn_f8_scale)."""
    out = torch.empty(4, dtype=torch.float32, device=a.device)          # [s_a, s_w, scratch, scratch]
    for i, t in enumerate((a, w)):
        c = t.shape[-1]
        lib().f8_scale(t.data_ptr(), t.numel() // c if t.is_contiguous() else t.numel() // c, c,
                       c if t.is_contiguous() else t.stride(-2), out[i:].data_ptr(), out[2 + i:].data_ptr(), _s())
    return out


class PreAct:
    """The output of a conv + train-mode BatchNorm + activation layer that was never materialised: ``y`` is the raw convolution
    output, the consumer forms act(scale*y + shift) where it loads its input (dcn_conv2d_fwd_pre / dcn_conv2d_bwd_weight_pre)."""
    __slots__ = ("y", "scale", "shift", "act", "slope")

    def __init__(self, y, scale, shift, act, slope):
        self.y, self.scale, self.shift, self.act, self.slope = y, scale, shift, act, slope

    shape = property(lambda self: self.y.shape)
    device = property(lambda self: self.y.device)

    def materialise(self, amax_out=None):
        return scale_act(self.y, self.scale, self.shift, self.act, self.slope, amax_out=amax_out)


PRE_ACT = True             # A/B switch: False = every BatchNorm + activation output is written by scale_act; "check" = written, but read
                           # with the abs-max bound the fused path uses (bit-identical to it: tests)


def pre_supported(n, h, wd, cin, cout, ksize, stride) -> bool:
    """Can BOTH readers of a layer's input (its forward and its weight gradient) form the activation of the layer in front themselves?"""
    return bool(PRE_ACT and _precision == "fp32" and lib().conv2d_pre_supported(n, h, wd, cin, cout, ksize, stride)
                and lib().conv2d_bwd_weight_pre_supported(n, h, wd, cin, cout, ksize, stride))


def bn_act_amax_bound(amax_y, scale, shift, slope):
    """abs-max word (a bound) of act(scale*y + shift) from the abs-max word of y."""
    out = amax_slot(amax_y.device)
    lib().bn_act_amax_bound(amax_y.data_ptr(), scale.data_ptr(), shift.data_ptr(), scale.numel(), float(slope), out.data_ptr(), _s())
    return out


def conv2d_fwd(x, w_ohwi, ksize, stride, scale=None, shift=None, act=ACT_NONE, slope=0.0,
               residual=None, out=None, want_stats=False, accumulate=False, amax_x=None, amax_w=None, amax_out=None,
               w_split_ready=None, w_b16=None):
    """x (N,H,W,Cin) NHWC, w_ohwi (Cout,k,k,Cin) [or (Cout,64) for the stem].  Returns (y, stats)
    where stats is the [rows][2][Cout] partial-sum buffer (None unless want_stats).
    amax_x / amax_w: abs-max words of the operands (computed here by a pass over the data when missing and the
    precision mode needs them); amax_out: word that receives the abs-max of what is stored.
    x may be a PreAct (pre_supported shapes, raw output only: no epilogue arguments); amax_x is then the word of the activation."""
    if isinstance(x, PreAct):
        if not (scale is None and shift is None and residual is None and act == ACT_NONE and not accumulate and amax_out is None
                and amax_x is not None and amax_w is not None):
            raise DcnError("conv2d_fwd: a PreAct input takes no epilogue and needs both abs-max words")
        _chk(x.y, "conv2d_fwd x")
        n, h, wd, cin = x.shape
        cout = w_ohwi.shape[0]
        ho, wo = conv_out_hw(h, wd, ksize, stride)
        if out is None:
            out = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        stats = None
        if want_stats:
            stats = torch.empty((lib().conv2d_stats_rows(n, h, wd, cout, ksize, stride), 2, cout), dtype=torch.float32, device=x.device)
        lib().conv2d_fwd_pre(x.y.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), n, h, wd, cin, cout, ksize, stride,
                             x.scale.data_ptr(), x.shift.data_ptr(), x.act, float(x.slope), out.stride(2), _p(stats),
                             amax_x.data_ptr(), amax_w.data_ptr(), _s())
        return out, stats
    _chk(x, "conv2d_fwd x")
    n, h, wd, cin = x.shape
    cout = w_ohwi.shape[0]
    ho, wo = conv_out_hw(h, wd, ksize, stride)
    if out is None:
        out = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    ldy = out.stride(2)
    stats = None
    if want_stats:
        rows = lib().conv2d_stats_rows(n, h, wd, cout, ksize, stride)
        stats = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
    f8 = None                        # (fp8 mode: the tiles derive their scales from the abs-max words below)
    wsplit = None
    if cin != 4:
        amax_x = _amax_or_pass(x, amax_x); amax_w = _amax_or_pass(w_ohwi, amax_w)
        if w_split_ready is not None and amax_w is not None and _precision == "fp32":
            wsplit = w_split_ready                                        # prepared for the whole network (FilterBanks)
        elif amax_w is not None and _precision == "fp32":
            wsplit = scratch(w_ohwi.numel() + 16, x.device, slot=5)       # the filter bank, split once per launch
    else:
        wsplit = scratch(27 * 32 + 16, x.device, slot=5)                  # the stem kernel's re-ordered filter bank
    ready = int(wsplit is not None and wsplit is w_split_ready)
    if w_b16 is not None and cin != 4 and _precision == "bf16":
        wsplit, ready = w_b16, 2                                          # the bank in bf16 (FilterBanks): the strip kernel's operand
    lib().conv2d_fwd(x.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), n, h, wd, cin, cout, ksize, stride,
                     _p(scale), _p(shift), act, float(slope), _p(residual),
                     0 if residual is None else residual.stride(2), ldy, _p(stats), int(accumulate), _p(f8),
                     _p(amax_x), _p(amax_w), _p(amax_out), _p(wsplit), ready, _s())
    return out, stats


def conv2d_bwd_data(dy, w_ohwi, in_hw, ksize, stride, out=None, accumulate=False, amax_dy=None, amax_w=None, wt_ready=None,
                    wt_b16=None, tap=None):
    """dy (N,Ho,Wo,Cout) (pixel stride may exceed Cout), w_ohwi (Cout,k,k,Cin) -> dx (N,H,W,Cin).
    wt_ready = (transposed fp32 bank, its split form | None) prepared by FilterBanks: nothing is converted here.
    tap = dict(y, mean, invstd, gamma, beta, act, slope) of the BatchNorm + activation whose output this convolution reads (and dx
    its complete gradient): returns (dx, partials | None) — the [rows][2][Cin] partial sums bn_act_bwd starts with, when the launch
    could form them (dcn_conv2d_bwd_data_tap), else None."""
    n, ho, wo, cout = dy.shape
    cin = w_ohwi.shape[3]
    h, wd = in_hw
    if out is None:
        out = torch.empty((n, h, wd, cin), dtype=torch.float32, device=dy.device)
    f8 = None
    amax_dy = _amax_or_pass(dy, amax_dy); amax_w = _amax_or_pass(w_ohwi, amax_w)
    if wt_ready is not None:
        wt, wts = wt_ready
        if _precision != "fp32":
            wts = None
    else:
        wt, wts = scratch(w_ohwi.numel() + 16, dy.device, slot=1), None
    ready = int(wt_ready is not None)
    if wt_ready is not None and wt_b16 is not None and _precision == "bf16":
        wts, ready = wt_b16, 2                                            # transposed bank in bf16 (FilterBanks)
    if tap is not None:
        # BatchNorm tap (csrc/nconv.hip): the partial sums of the backward of the layer in front, formed in this launch's epilogue
        import ctypes
        cap = lib().conv2d_bwd_data_tap_rows(n, h, wd, cin, cout, ksize, stride)
        part = torch.empty((max(cap, 1), 2, cin), dtype=torch.float32, device=dy.device)
        rows = ctypes.c_int(0)
        lib().conv2d_bwd_data_tap(dy.data_ptr(), dy.stride(2), w_ohwi.data_ptr(), wt.data_ptr(), out.data_ptr(),
                                  n, h, wd, cin, cout, ksize, stride, int(accumulate), _p(f8), _p(amax_dy), _p(amax_w),
                                  ready, _p(wts), tap["y"].data_ptr() if cap else 0, tap["mean"].data_ptr(), tap["invstd"].data_ptr(),
                                  _p(tap.get("gamma")), _p(tap.get("beta")), int(tap["act"]), float(tap["slope"]),
                                  part.data_ptr() if cap else 0, cap, ctypes.addressof(rows), _s())
        return out, (part[:rows.value] if rows.value > 0 else None)
    lib().conv2d_bwd_data(dy.data_ptr(), dy.stride(2), w_ohwi.data_ptr(), wt.data_ptr(), out.data_ptr(),
                          n, h, wd, cin, cout, ksize, stride, int(accumulate), _p(f8), _p(amax_dy), _p(amax_w),
                          ready, _p(wts), _s())
    return out


def conv2d_fwd_b16(x, w16, cout, ksize, stride, scale=None, shift=None, act=ACT_NONE, slope=0.0, residual=None, out=None,
                   want_stats=False, accumulate=False, out_f32=False):
    """bf16 storage: x (N,H,W,Cin) bf16 NHWC, w16 the bf16 bank [Cout][k*k*Cin] (FilterBanks "b16").  Returns (y, stats): y bf16
    (fp32 with out_f32), stats [rows][2][Cout] fp32 partial sums of the raw result as stored (None unless want_stats)."""
    _chk16(x, "conv2d_fwd_b16 x")
    n, h, wd, cin = x.shape
    if not (w16.is_cuda and w16.dtype == torch.bfloat16 and w16.is_contiguous() and w16.numel() == cout * ksize * ksize * cin):
        raise ValueError("conv2d_fwd_b16: w16 must be the contiguous bf16 bank [Cout][k*k*Cin]")
    ho, wo = conv_out_hw(h, wd, ksize, stride)
    if out is None:
        out = torch.empty((n, ho, wo, cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    if out.dtype != (torch.float32 if out_f32 else torch.bfloat16) or out.stride(3) != 1:
        raise ValueError("conv2d_fwd_b16: out dtype / layout")
    if residual is not None:
        _rows16(residual, "conv2d_fwd_b16 residual")
        if not _b16(residual):
            raise ValueError("conv2d_fwd_b16: residual must be bf16")
    stats = None
    if want_stats:
        stats = torch.empty((lib().conv2d_stats_rows_b16(n, h, wd, cout, ksize, stride), 2, cout), dtype=torch.float32, device=x.device)
    lib().conv2d_fwd_b16(x.data_ptr(), w16.data_ptr(), out.data_ptr(), int(out_f32), n, h, wd, cin, cout, ksize, stride, _p(scale), _p(shift),
                         act, float(slope), _p(residual), 0 if residual is None else residual.stride(2), out.stride(2), _p(stats),
                         int(accumulate), _s())
    return out, stats


def conv2d_bwd_data_b16(dy, wt16, in_hw, cin, ksize, stride, out=None, accumulate=False, tap=None, out_f32=False):
    """bf16 storage: dy (N,Ho,Wo,Cout) bf16 (pixel stride may exceed Cout), wt16 the transposed bf16 bank [Cin][k*k*Cout]
    (FilterBanks "tb16") -> dx (N,H,W,Cin) bf16 (fp32 with out_f32).  tap: as conv2d_bwd_data (y bf16); returns (dx, partials | None)."""
    import ctypes
    n, ho, wo, cout = dy.shape
    h, wd = in_hw
    _rows16(dy, "conv2d_bwd_data_b16 dy")
    if not _b16(dy) or not (wt16.dtype == torch.bfloat16 and wt16.is_contiguous() and wt16.numel() == cin * ksize * ksize * cout):
        raise ValueError("conv2d_bwd_data_b16: dy must be bf16, wt16 the contiguous bf16 bank [Cin][k*k*Cout]")
    if out is None:
        out = torch.empty((n, h, wd, cin), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dy.device)
    if out.dtype != (torch.float32 if out_f32 else torch.bfloat16) or not out.is_contiguous():
        raise ValueError("conv2d_bwd_data_b16: out dtype / layout")
    part = None
    rows = ctypes.c_int(0)
    cap = 0
    if tap is not None and stride == 1 and _b16(tap["y"]) and tap["y"].is_contiguous():
        cap = lib().conv2d_stats_rows_b16(n, h, wd, cin, ksize, 1)        # M-tiles of the launch (rows = n*h*wd, filters = cin, k*k taps: stride 1 keeps h x wd)
        part = torch.empty((max(cap, 1), 2, cin), dtype=torch.float32, device=dy.device)
    lib().conv2d_bwd_data_b16(dy.data_ptr(), dy.stride(2), wt16.data_ptr(), out.data_ptr(), int(out_f32), n, h, wd, cin, cout, ksize, stride,
                              int(accumulate), tap["y"].data_ptr() if cap else 0, tap["mean"].data_ptr() if cap else 0,
                              tap["invstd"].data_ptr() if cap else 0, _p(tap.get("gamma")) if cap else 0, _p(tap.get("beta")) if cap else 0,
                              int(tap["act"]) if cap else 0, float(tap["slope"]) if cap else 0.0, part.data_ptr() if cap else 0, cap,
                              ctypes.addressof(rows), _s())
    if tap is not None:
        return out, (part[:rows.value] if rows.value > 0 else None)
    return out


# ---- fp8 storage (configs[4]): e4m3 bytes + one e8m0 scale per row -----------------------------------------------------------------
def quant_rows_e4m3(x: torch.Tensor):
    """x (..., c) bf16 rows (c % 8 == 0, last dim contiguous) -> (q uint8 same shape: OCP e4m3 bytes, scales uint8 [rows]: e8m0, one per
    row): q = e4m3(x * 2^-e), e = floor(log2(max|row|)) - 8 (csrc/b16.hip)."""
    _rows16(x, "quant_rows_e4m3 x")
    if not _b16(x):
        raise ValueError("quant_rows_e4m3: x must be bf16")
    c = x.shape[-1]
    rows = x.numel() // c
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.uint8, device=x.device)
    lib().quant_rows_e4m3(x.data_ptr(), x.stride(-2) if x.dim() > 1 else c, rows, c, q.data_ptr(), c, sc.data_ptr(), _s())
    return q, sc


def bank_q8(bank, key: str, b16: torch.Tensor, rows: int):
    """The e4m3 form of a filter bank for the fp8-storage mode: the one FilterBanks.refresh made this step (``key``: "q8" forward bank,
    "tq8" transposed bank of the data gradient), else a quantisation pass over the bf16 bank now."""
    q = bank.get(key) if isinstance(bank, dict) else None
    return q if q is not None else quant_rows_e4m3(b16.view(rows, -1))


def conv2d_fwd_f8(x8, xs, w8, ws, cout, ksize, stride, scale=None, shift=None, act=ACT_NONE, slope=0.0, residual=None, want_stats=False,
                  out_f32=False):
    """fp8 storage: x8 (N,H,W,Cin) uint8 e4m3 with xs [N*H*W] e8m0, w8 [Cout][k*k*Cin] e4m3 with ws [Cout] (quant_rows_e4m3 of the bf16
    tensors).  Returns (y bf16 | fp32, stats | None) as conv2d_fwd_b16."""
    n, h, wd, cin = x8.shape
    if not (x8.dtype == torch.uint8 and x8.is_contiguous() and w8.dtype == torch.uint8 and w8.is_contiguous() and w8.numel() == cout * ksize * ksize * cin
            and xs.dtype == torch.uint8 and xs.numel() == n * h * wd and ws.dtype == torch.uint8 and ws.numel() == cout):
        raise ValueError("conv2d_fwd_f8: x8 / w8 must be contiguous uint8 e4m3 tensors with one e8m0 byte per pixel / per filter")
    ho, wo = conv_out_hw(h, wd, ksize, stride)
    out = torch.empty((n, ho, wo, cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x8.device)
    if residual is not None and not (_b16(residual) and residual.is_contiguous()):
        raise ValueError("conv2d_fwd_f8: residual must be a contiguous bf16 tensor")
    stats = None
    if want_stats:
        stats = torch.empty((lib().conv2d_stats_rows_f8(n, h, wd, cout, ksize, stride), 2, cout), dtype=torch.float32, device=x8.device)
    lib().conv2d_fwd_f8(x8.data_ptr(), xs.data_ptr(), w8.data_ptr(), ws.data_ptr(), out.data_ptr(), int(out_f32), n, h, wd, cin, cout, ksize, stride,
                        _p(scale), _p(shift), act, float(slope), _p(residual), 0, cout, _p(stats), 0, _s())
    return out, stats


def conv2d_bwd_data_f8(dy8, dys, wt8, wts, in_hw, cin, ksize, stride, out=None, accumulate=False, tap=None, out_f32=False):
    """fp8 storage: dy8 (N,Ho,Wo,Cout) uint8 e4m3 with dys per pixel, wt8 the transposed bank [Cin][k*k*Cout] e4m3 with wts per row -> dx
    (N,H,W,Cin) bf16 (fp32 with out_f32).  tap: as conv2d_bwd_data_b16; returns (dx, partials | None) then."""
    import ctypes
    n, ho, wo, cout = dy8.shape
    h, wd = in_hw
    if not (dy8.dtype == torch.uint8 and dy8.is_contiguous() and wt8.dtype == torch.uint8 and wt8.is_contiguous()
            and wt8.numel() == cin * ksize * ksize * cout and dys.numel() == n * ho * wo and wts.numel() == cin):
        raise ValueError("conv2d_bwd_data_f8: dy8 / wt8 must be contiguous uint8 e4m3 tensors with one e8m0 byte per pixel / per bank row")
    if out is None:
        out = torch.empty((n, h, wd, cin), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dy8.device)
    if out.dtype != (torch.float32 if out_f32 else torch.bfloat16) or not out.is_contiguous():
        raise ValueError("conv2d_bwd_data_f8: out dtype / layout")
    part = None
    rows = ctypes.c_int(0)
    cap = 0
    if tap is not None and stride == 1 and _b16(tap["y"]) and tap["y"].is_contiguous():
        cap = lib().conv2d_stats_rows_f8(n, h, wd, cin, ksize, 1)
        part = torch.empty((max(cap, 1), 2, cin), dtype=torch.float32, device=dy8.device)
    lib().conv2d_bwd_data_f8(dy8.data_ptr(), dys.data_ptr(), wt8.data_ptr(), wts.data_ptr(), out.data_ptr(), int(out_f32), n, h, wd, cin, cout, ksize,
                             stride, int(accumulate), tap["y"].data_ptr() if cap else 0, tap["mean"].data_ptr() if cap else 0,
                             tap["invstd"].data_ptr() if cap else 0, _p(tap.get("gamma")) if cap else 0, _p(tap.get("beta")) if cap else 0,
                             int(tap["act"]) if cap else 0, float(tap["slope"]) if cap else 0.0, part.data_ptr() if cap else 0, cap,
                             ctypes.addressof(rows), _s())
    if tap is not None:
        return out, (part[:rows.value] if rows.value > 0 else None)
    return out


def conv2d_bwd_weight_b16(x, dy, ksize, stride, slot: int = 0):
    """bf16 storage: x (N,H,W,Cin), dy (N,Ho,Wo,Cout) bf16 -> dw OHWI (Cout,k,k,Cin) fp32."""
    n, h, wd, cin = x.shape
    cout = dy.shape[3]
    _rows16(x, "conv2d_bwd_weight_b16 x"); _rows16(dy, "conv2d_bwd_weight_b16 dy")
    if not (_b16(x) and _b16(dy)):
        raise ValueError("conv2d_bwd_weight_b16: bf16 tensors only")
    dw = torch.empty((cout, ksize, ksize, cin), dtype=torch.float32, device=x.device)
    if STEP_ABL & 1:
        return dw
    nws = lib().conv2d_bwd_weight_ws_b16(n, h, wd, cin, cout, ksize, stride)
    ws = scratch(nws, x.device, slot=slot) if nws > 0 else None
    cnt = slab_counters(x.device, slot) if nws > 0 else None
    geom = conv_geom(x.device, n, h, wd, ksize, stride)
    lib().conv2d_bwd_weight_b16(x.data_ptr(), x.stride(2), dy.data_ptr(), dy.stride(2), dw.data_ptr(), _p(ws), _p(cnt), geom.data_ptr(),
                                n, h, wd, cin, cout, ksize, stride, _s())
    return dw


FILTER_BANKS = True      # A/B switch: False = per-layer transposes / abs-max / pre-split again


class FilterBanks:
    """The filter banks of a list of convolutions in every form a step needs, refreshed by ONE dcn_prepare_filters call
    (three launches) instead of four small kernels per layer and direction: OHWI fp32, OHWI split into f16 pieces, the
    channel-transposed bank of the data gradient (fp32 and split) and the bank's abs-max word.
    ``weights``: OIHW parameters with Cin and Cout multiples of 32 (others keep the per-layer path: ``get`` returns None)."""

    def __init__(self, weights, device):
        import struct
        self.device = torch.device(device)
        self.items = {}
        pairs = weights.items() if isinstance(weights, dict) else enumerate(weights)
        ok = [(i, w) for i, w in pairs if w is not None and w.dim() == 4 and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0
              and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()]
        self.amax = torch.zeros(max(1, len(ok)) * AMAX_WORDS, dtype=torch.int32, device=self.device)
        rec = lib().filter_job_bytes()
        assert rec == 8 * 8 + 6 * 4, rec
        blob = bytearray()
        blk = ablk = 0
        self._ptrs = []
        for j, (i, w) in enumerate(ok):
            co, ci, kh, kw = w.shape
            T = kh * kw
            numel = w.numel()
            ohwi = torch.empty((co, kh, kw, ci), dtype=torch.float32, device=self.device) if T > 1 else None
            split = torch.empty(numel + 16, dtype=torch.float32, device=self.device)
            t = torch.empty(numel + 16, dtype=torch.float32, device=self.device)
            tsplit = torch.empty(numel + 16, dtype=torch.float32, device=self.device)
            am = self.amax[j * AMAX_WORDS:(j + 1) * AMAX_WORDS]
            b16 = torch.empty(numel, dtype=torch.bfloat16, device=self.device)          # the banks of the bf16-operand mode
            tb16 = torch.empty(numel, dtype=torch.bfloat16, device=self.device)
            self.items[i] = dict(ohwi=ohwi, split=split, t=t, tsplit=tsplit, amax=am, b16=b16, tb16=tb16, shape=(co, ci, kh, kw))
            blob += struct.pack("<8Q6i", w.data_ptr(), 0 if ohwi is None else ohwi.data_ptr(), split.data_ptr(), t.data_ptr(),
                                tsplit.data_ptr(), am.data_ptr(), b16.data_ptr(), tb16.data_ptr(), co, ci, T, blk, ablk, 0)
            self._ptrs.append((i, w.data_ptr()))
            blk += T * (co // 32) * (ci // 32); ablk += (numel + 4095) // 4096
        self.njobs, self.blocks, self.ablocks = len(ok), blk, ablk
        self._q8_jobs = None; self._q8_key = None
        self.jobs = torch.frombuffer(blob, dtype=torch.uint8).clone().to(self.device) if ok else None

    def valid_for(self, weights) -> bool:
        """The job table holds raw parameter addresses: it is stale once a parameter moved (``.to()``, a new tensor)."""
        return all(weights.get(i) is not None and weights[i].data_ptr() == ptr for i, ptr in self._ptrs) if isinstance(weights, dict) \
            else all(weights[i] is not None and weights[i].data_ptr() == ptr for i, ptr in self._ptrs)

    def refresh(self):
        if self.njobs:
            lib().prepare_filters(self.jobs.data_ptr(), self.njobs, self.blocks, self.ablocks, self.amax.data_ptr(),
                                  self.amax.numel(), _s())
            if storage_f8():
                self._refresh_q8()

    def _refresh_q8(self):
        """fp8 storage: the e4m3 forms of the banks the mode reads (3x3 layers: forward ``q8`` = (bytes [Cout][k*k*Cin], scales [Cout]),
        data gradient ``tq8`` over the transposed bank) in ONE launch behind the refresh, instead of a quant_rows_e4m3 launch per layer
        in front of its forward convolution and another in front of its data gradient (77 launches on the step's critical chain)."""
        import struct
        if self._q8_jobs is None or self._q8_key != F8_MIN_K:
            rec = lib().quant_job_bytes()
            assert rec == 3 * 8 + 4 * 4, rec
            blob = bytearray(); blk = 0; elems = 0; n = 0
            for it in self.items.values():
                co, ci, kh, kw = it["shape"]
                it.pop("q8", None); it.pop("tq8", None)
                for key, src, rows, ok in (("q8", it["b16"], co, f8_takes(ci, co, kh)), ("tq8", it["tb16"], ci, f8_takes(co, ci, kh))):
                    if not ok:
                        continue
                    c = src.numel() // rows
                    q = torch.empty((rows, c), dtype=torch.uint8, device=self.device)
                    sc = torch.empty(rows, dtype=torch.uint8, device=self.device)
                    it[key] = (q, sc)
                    blob += struct.pack("<3Q4i", src.data_ptr(), q.data_ptr(), sc.data_ptr(), rows, c, blk, 0)
                    blk += (rows + 3) // 4; elems += rows * c; n += 1
            self._q8_jobs = (torch.frombuffer(blob, dtype=torch.uint8).clone().to(self.device), n, blk, elems) if n else (None, 0, 0, 0)
            self._q8_key = F8_MIN_K
        jobs, n, blk, elems = self._q8_jobs
        if n:
            lib().quant_rows_e4m3_batched(jobs.data_ptr(), n, blk, elems, _s())

    def get(self, i, w):
        """dict(ohwi (Cout,k,k,Cin), split, t, tsplit, amax) of weight ``i``, or None when it is not in the table."""
        it = self.items.get(i)
        if it is None:
            return None
        if it["ohwi"] is None:
            co, ci, kh, kw = it["shape"]
            it = dict(it); it["ohwi"] = w.detach().view(co, 1, 1, ci)       # a 1x1 bank is the same bytes in OIHW and OHWI
        return it


_geom = {}


def conv_geom(device, n: int, h: int, wd: int, ksize: int, stride: int) -> torch.Tensor:
    """The gather table of a conv geometry (dcn_conv2d_geom), built on first use and kept per device."""
    key = (torch.device(device).index, n, h, wd, ksize, stride)
    t = _geom.get(key)
    if t is None:
        t = torch.empty(lib().conv2d_geom_size(n, h, wd, ksize, stride), dtype=torch.int32, device=device)
        lib().conv2d_geom(t.data_ptr(), n, h, wd, ksize, stride, _s())
        torch.cuda.current_stream().synchronize()      # once per geometry: other streams may use it right away
        _geom[key] = t
    return t


STEP_ABL = 0           # timing-only ablations of a whole step (results WRONG; bench.py --schedule-tunes "ops.STEP_ABL=1=0"): bit 1 = no weight-gradient
                       # launch at all (dw comes back uninitialised) — what the weight-gradient queue costs the step's wall time; bit 2 = no
                       # persistent BiLSTM backward (it holds 64 KB of LDS on half the CUs for 2.3 ms beside the backbone's backward)


def conv2d_bwd_weight(x, dy, ksize, stride, cout=None, slot: int = 0, amax_x=None, amax_dy=None):
    """x (N,H,W,Cin) NHWC (Cin == 4: stem), dy (N,Ho,Wo,Cout) -> dw OHWI (Cout,k,k,Cin) [(Cout,64) stem].
    ``slot`` selects the scratch buffer for the split-K slabs (a side stream must not share slot 0).
    x may be a PreAct (see conv2d_fwd); amax_x is then the word of the activation."""
    if _b16(dy) and not isinstance(x, PreAct):
        return conv2d_bwd_weight_b16(x, dy, ksize, stride, slot=slot)
    n, h, wd, cin = x.shape
    cout = dy.shape[3] if cout is None else cout
    dw = torch.empty((cout, 64) if cin == 4 else (cout, ksize, ksize, cin), dtype=torch.float32, device=x.device)
    if STEP_ABL & 1:
        return dw
    nws = lib().conv2d_bwd_weight_ws(n, h, wd, cin, cout, ksize, stride)
    ws = scratch(nws, x.device, slot=slot) if nws > 0 else None
    cnt = slab_counters(x.device, slot) if nws > 0 else None
    if isinstance(x, PreAct):
        if amax_x is None:
            raise DcnError("conv2d_bwd_weight: a PreAct input needs the abs-max word of the activation")
        amax_dy = _amax_or_pass(dy, amax_dy)
        lib().conv2d_bwd_weight_pre(x.y.data_ptr(), x.y.stride(2), dy.data_ptr(), dy.stride(2), dw.data_ptr(), _p(ws), _p(cnt),
                                    n, h, wd, cin, cout, ksize, stride, x.scale.data_ptr(), x.shift.data_ptr(), x.act, float(x.slope),
                                    amax_x.data_ptr(), amax_dy.data_ptr(), _s())
        return dw
    geom = conv_geom(x.device, n, h, wd, ksize, stride)
    # the kernels with a split mode: the 128-wide weight-gradient tiles and the nine-tap kernel of the 32 -> 64 3x3 layers
    if (cin >= 64 and cout >= 64 and (cin >= 128 or cout >= 128)) or (cin == 32 and cout == 64 and ksize == 3):      # (64 -> 128 is in the first)
        amax_x = _amax_or_pass(x, amax_x); amax_dy = _amax_or_pass(dy, amax_dy)
    lib().conv2d_bwd_weight(x.data_ptr(), x.stride(2), dy.data_ptr(), dy.stride(2), dw.data_ptr(), _p(ws), _p(cnt), geom.data_ptr(),
                            n, h, wd, cin, cout, ksize, stride, _p(amax_x), _p(amax_dy), _s())
    return dw


# ---- side stream for the weight gradient --------------------------------------------------------------
# dW and dX of a layer both depend only on dY, so they are launched on two streams: the partially filled
# last round of one kernel's grid is topped up with workgroups of the other (256 CUs x 2 slots).
_side = {}


def side_stream(device) -> "torch.cuda.Stream":
    """The weight-gradient companion of the current stream (one per calling stream, so the per-scale
    head branches do not serialise on each other's weight gradients)."""
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream, SIDE_PRIORITY)
    if key not in _side:
        if SIDE_PRIORITY:
            with torch.cuda.device(device):
                handle = lib().stream_create(SIDE_PRIORITY)
            if not handle:
                raise DcnError("dcn_stream_create failed")
            _side[key] = torch.cuda.ExternalStream(handle, device=device)
        else:
            _side[key] = torch.cuda.Stream(device=device)
    return _side[key]


SIDE_PRIORITY = 0      # 0: a normal torch stream; +1 (lowest dispatch priority, dcn_stream_create) measured 20 % slower
WGRAD_SIDE = True      # A/B switch: False runs the weight gradient on the caller's stream
WGRAD_AFTER_DGRAD = True    # the side stream also waits for the layer's DATA gradient (queued first: the critical chain), so a weight
                            # gradient starts beside the NEXT layer's HBM-bound BatchNorm passes instead of beside its own data gradient:
                            # 110.2-112.0 vs 111.0-112.9 ms per replayed step in three same-process comparisons (bench.py --schedules;
                            # False = round 2's order; a lowest-priority side stream: +-0)


WGRAD_HELD = True      # A/B switch (darknet._run_backward): a layer's weight gradient is LAUNCHED behind the next layer's BatchNorm passes (its
                       # dependency stays the event recorded behind its own data gradient).  Same DAG; in a captured step the main chain
                       # then is the first dependent of every data gradient and the graph executor keeps it on ONE queue, the weight
                       # gradients on another (without it the main chain hops queues layer by layer and meets the weight-gradient queue —
                       # and whatever else runs beside it — every fourth layer: tools/graph_sched.py)


class HeldWgrad:
    """A weight gradient on the side stream whose launch is held back: the constructor records (on the current stream) the event it
    depends on, issue() launches it.  issue() must run with the same current stream."""

    def __init__(self, x, dy, ksize, stride, wshape, amax_x=None, amax_dy=None):
        self.args = (x, dy, ksize, stride, wshape, amax_x, amax_dy)
        self.main = torch.cuda.current_stream()
        self.side = side_stream(x.device)            # the companion of the stream that produced dy, whoever launches the gradient later
        self.event = torch.cuda.Event()
        self.event.record(self.main)

    def issue(self):
        x, dy, ksize, stride, wshape, amax_x, amax_dy = self.args
        self.args = None
        self.side.wait_event(self.event)             # dy (and x) are produced on the main stream, in front of the event
        return _wgrad_side_launch(self.main, self.side, x, dy, ksize, stride, wshape, amax_x, amax_dy)


def _wgrad_side_launch(main, side, x, dy, ksize, stride, wshape, amax_x, amax_dy):
    with torch.cuda.stream(side):
        dw = conv2d_bwd_weight(x, dy, ksize, stride, slot=3, amax_x=amax_x, amax_dy=amax_dy)
        if wshape[0] != dw.shape[0]:
            dw = dw[:wshape[0]].contiguous()
        out = weight_grad_to_oihw(dw, wshape)
    for t in ((x.y, x.scale, x.shift) if isinstance(x, PreAct) else (x,)) + (dy,):
        t.record_stream(side)                    # keep the allocator from recycling them under the side kernels
    out.record_stream(main)
    return out


def wgrad_on_side(x, dy, ksize, stride, wshape, amax_x=None, amax_dy=None):
    """Launch the weight gradient (+ its OHWI->OIHW conversion) on the side stream.  Returns the OIHW
    gradient; the caller must make the main stream wait for side_stream() before the result is consumed."""
    if not WGRAD_SIDE:
        dw = conv2d_bwd_weight(x, dy, ksize, stride, amax_x=amax_x, amax_dy=amax_dy)
        if wshape[0] != dw.shape[0]:
            dw = dw[:wshape[0]].contiguous()
        return weight_grad_to_oihw(dw, wshape)
    main = torch.cuda.current_stream()
    side = side_stream(x.device)
    side.wait_stream(main)                       # dy (and x) are produced on the main stream
    return _wgrad_side_launch(main, side, x, dy, ksize, stride, wshape, amax_x, amax_dy)


WGRAD_DIRECT = False   # set by a step driver that calls finish_wgrads() behind loss.backward() (graph.GraphedTrainStep).  The head's conv blocks
                       # (functions.ConvBNAct / ConvBias) then do NOT hand their weight gradient to autograd — that needs the main stream to
                       # wait for the side stream in every block: dgrad -> wgrad -> next block, nothing beside anything — but add it to
                       # the parameter's .grad themselves, ON the side stream, launched behind the next block's BatchNorm passes like the
                       # backbone's (WGRAD_HELD); the main stream joins once, in finish_wgrads().  Same sums into the same buffers.
                       # (Parameter hooks do not fire for these gradients: drivers whose reducer hangs on hooks leave it off.)
HEAD_WGRAD_DIRECT = True    # A/B switch: graph.GraphedTrainStep sets WGRAD_DIRECT for its step
_held_direct: list = []


def hold_wgrad_into(param, x, dy, ksize, stride, wshape, amax_x=None, amax_dy=None) -> None:
    """The weight gradient of (x, dy), to be added to param.grad on the side stream; launched by the next release_held_wgrads()."""
    _held_direct.append((HeldWgrad(x, dy, ksize, stride, wshape, amax_x=amax_x, amax_dy=amax_dy), param))


def reset_held_wgrads() -> None:
    """Drop what a backward that did not finish (an exception between hold and release) left behind."""
    _held_direct.clear()


def release_held_wgrads() -> None:
    while _held_direct:
        hw, param = _held_direct.pop(0)
        out = hw.issue()
        with torch.cuda.stream(hw.side):
            if param.grad is None:
                param.grad = out.view_as(param)
            else:
                param.grad.add_(out.view_as(param))


def finish_wgrads(device) -> None:
    """Launch what is still held and make the current stream wait for every weight-gradient stream of the device."""
    release_held_wgrads()
    cur = torch.cuda.current_stream(device)
    idx = torch.device(device).index
    for key, st in _side.items():
        if key[0] == idx and st is not cur:
            cur.wait_stream(st)


def join_side(device) -> None:
    if WGRAD_SIDE:
        torch.cuda.current_stream().wait_stream(side_stream(device))


# ---- batch norm ------------------------------------------------------------------------------------
def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var):
    rows, _, c = stats.shape
    dev = stats.device
    res = torch.empty((4, c), dtype=torch.float32, device=dev)      # mean, invstd, scale, shift
    ws = scratch(lib().bn_ws(c), dev, slot=2)
    lib().bn_finalize(stats.data_ptr(), rows, c, int(count), _p(gamma), _p(beta), float(eps), float(momentum),
                      _p(running_mean), _p(running_var), res[0].data_ptr(), res[1].data_ptr(),
                      res[2].data_ptr(), res[3].data_ptr(), ws.data_ptr(), _s())
    return res


def bn_fold(gamma, beta, running_mean, running_var, eps=1e-5):
    c = gamma.numel()
    res = torch.empty((2, c), dtype=torch.float32, device=gamma.device)
    lib().bn_fold(gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), float(eps), c,
                  res[0].data_ptr(), res[1].data_ptr(), _s())
    return res


def channel_stats(x2d):
    """x2d [rows][c] contiguous -> partials [R][2][c]."""
    rows, c = x2d.shape
    r = lib().channel_stats_rows(rows)
    stats = torch.empty((r, 2, c), dtype=torch.float32, device=x2d.device)
    lib().channel_stats(x2d.data_ptr(), rows, c, c, stats.data_ptr(), _s())
    return stats


def quant_of(x: torch.Tensor):
    """(q8, scales) of a bf16 tensor for the fp8-storage convolutions: the copy its producing pass wrote beside it (scale_act / bn_act_bwd
    with quant=True), else a quantisation pass now."""
    q = getattr(x, "_dcn_q8", None)
    return q if q is not None else quant_rows_e4m3(x)


def scale_act(y, scale, shift, act, slope, residual=None, out=None, amax_out=None, out_b16=False, out_f32=False, quant=False):
    """quant (fp8 storage): also write the e4m3 copy of the bf16 result, attached to it as ``out._dcn_q8 = (q8, scales)`` (quant_of)."""
    c = y.shape[-1]
    rows = y.numel() // c
    if quant and _b16(y) and out is None and not out_f32 and y.is_contiguous() and lib().quant_fusable(c) and (residual is None or _b16(residual)):
        out = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
        q8 = torch.empty(y.shape, dtype=torch.uint8, device=y.device)
        qs = torch.empty(rows, dtype=torch.uint8, device=y.device)
        if residual is not None:
            _rows16(residual, "scale_act residual")
        lib().scale_act_b16_q(y.data_ptr(), _p(scale), _p(shift), act, float(slope), _p(residual), 0 if residual is None else residual.stride(-2),
                              out.data_ptr(), rows, c, q8.data_ptr(), qs.data_ptr(), _s())
        out._dcn_q8 = (q8, qs)
        return out
    if _b16(y) or out_b16 or _b16(out):
        # bf16 storage (csrc/b16.hip): y bf16 (fp32 for the stem's raw output), residual bf16, out bf16 (fp32 with out_f32: the boundary
        # to an fp32 consumer)
        if out is None:
            out = torch.empty(y.shape, dtype=torch.float32 if out_f32 else torch.bfloat16, device=y.device)
        if not y.is_contiguous() or (residual is not None and not _b16(residual)):
            raise ValueError("scale_act (bf16 storage): contiguous y, bf16 residual")
        _rows16(out, "scale_act out")
        if residual is not None:
            _rows16(residual, "scale_act residual")
        lib().scale_act_b16(y.data_ptr(), int(not _b16(y)), _p(scale), _p(shift), act, float(slope), _p(residual),
                            0 if residual is None else residual.stride(-2), out.data_ptr(), int(not _b16(out)), rows, c, out.stride(-2), _s())
        return out
    if out is None:
        out = torch.empty_like(y)
    _chk(y, "scale_act y"); _rows(out, "scale_act out")
    lib().scale_act(y.data_ptr(), _p(scale), _p(shift), act, float(slope), _p(residual), out.data_ptr(), rows, c,
                    out.stride(-2), _p(amax_out), _s())
    return out


def _bn_bwd_partials(y, dout, mean, invstd, gamma, beta, act, slope, part):
    """(partials, rows): the given ones (a conv2d_bwd_data tap) or those of a dcn_bn_act_bwd_reduce pass."""
    c = y.shape[-1]
    rows = y.numel() // c
    if part is not None:
        return part, part.shape[0]
    if _b16(dout) or _b16(y):
        r = lib().bn_act_bwd_reduce_rows_b16(rows)
        part = scratch(r * 2 * c, y.device, slot=0)
        lib().bn_act_bwd_reduce_b16(y.data_ptr(), int(not _b16(y)), dout.data_ptr(), int(not _b16(dout)), dout.stride(-2), mean.data_ptr(), invstd.data_ptr(),
                                    _p(gamma), _p(beta), act, float(slope), rows, c, part.data_ptr(), _s())
        return part, r
    r = lib().channel_stats_rows(rows)
    part = scratch(r * 2 * c, y.device, slot=0)
    lib().bn_act_bwd_reduce(y.data_ptr(), dout.data_ptr(), dout.stride(-2), mean.data_ptr(), invstd.data_ptr(), _p(gamma), _p(beta),
                            act, float(slope), rows, c, part.data_ptr(), _s())
    return part, r


def bn_act_bwd(y, dout, mean, invstd, gamma, beta, act, slope, amax_out=None, part=None, quant=False):
    """Returns (dy, dgamma, dbeta) for out = act(gamma*(y-mean)*invstd+beta) with batch statistics; amax_out: word that
    receives the abs-max of dy (the A operand of the data / weight gradient GEMMs that follow); part: partial sums already
    formed by the launch that produced dout (conv2d_bwd_data(tap=...))."""
    c = y.shape[-1]
    rows = y.numel() // c
    dev = y.device
    lddo = dout.stride(-2)
    part, r = _bn_bwd_partials(y, dout, mean, invstd, gamma, beta, act, slope, part)
    sums = torch.empty((2, c), dtype=torch.float32, device=dev)
    ws = scratch(lib().bn_ws(c), dev, slot=2)
    lib().bn_bwd_sums(part.data_ptr(), r, c, sums.data_ptr(), ws.data_ptr(), _s())
    if _b16(dout) or _b16(y):          # bf16 storage: dy bf16 (y bf16, or the stem's fp32 raw output; dout bf16, or fp32 at the boundary)
        dy = torch.empty(y.shape, dtype=torch.bfloat16, device=dev)
        if quant and _b16(y) and _b16(dout) and lib().quant_fusable(c):       # fp8 storage: the e4m3 copy of dy from the same pass (quant_of)
            q8 = torch.empty(y.shape, dtype=torch.uint8, device=dev)
            qs = torch.empty(rows, dtype=torch.uint8, device=dev)
            lib().bn_act_bwd_apply_b16_q(y.data_ptr(), dout.data_ptr(), lddo, mean.data_ptr(), invstd.data_ptr(), _p(gamma), _p(beta), act, float(slope),
                                         sums.data_ptr(), rows, rows, c, dy.data_ptr(), q8.data_ptr(), qs.data_ptr(), _s())
            dy._dcn_q8 = (q8, qs)
            return dy, sums[1], sums[0]
        lib().bn_act_bwd_apply_b16(y.data_ptr(), int(not _b16(y)), dout.data_ptr(), int(not _b16(dout)), lddo, mean.data_ptr(), invstd.data_ptr(), _p(gamma),
                                   _p(beta), act, float(slope), sums.data_ptr(), rows, rows, c, dy.data_ptr(), _s())
        return dy, sums[1], sums[0]
    dy = torch.empty_like(y)
    lib().bn_act_bwd_apply(y.data_ptr(), dout.data_ptr(), lddo, mean.data_ptr(), invstd.data_ptr(), _p(gamma), _p(beta),
                           act, float(slope), sums.data_ptr(), rows, rows, c, dy.data_ptr(), _p(amax_out), _s())
    return dy, sums[1], sums[0]


def stem_bwd_weight_bn(x, y, dout, mean, invstd, gamma, beta, act, slope, part=None):
    """The stem's weight gradient with its BatchNorm + activation backward applied on the fly (csrc/stem.hip): x (N,H,W,4),
    y (N,H,W,32) the raw convolution output, dout the gradient w.r.t. act(bn(y)).  Returns (dw (32, 64) in the c4 layout of
    conv2d_bwd_weight, dgamma, dbeta) — what bn_act_bwd + conv2d_bwd_weight return, without writing dy."""
    n, h, wd, c = y.shape
    rows = n * h * wd
    dev = y.device
    lddo = dout.stride(-2)
    part, r = _bn_bwd_partials(y, dout, mean, invstd, gamma, beta, act, slope, part)
    sums = torch.empty((2, c), dtype=torch.float32, device=dev)
    ws = scratch(lib().bn_ws(c), dev, slot=2)
    lib().bn_bwd_sums(part.data_ptr(), r, c, sums.data_ptr(), ws.data_ptr(), _s())
    dw = torch.empty((c, 64), dtype=torch.float32, device=dev)
    slabs = scratch(lib().stem_bwd_weight_bn_ws(n, h, wd), dev, slot=0)          # (slot 0's partials have been reduced by then; slot 3 belongs to the side stream)
    fn = lib().stem_bwd_weight_bn_b16 if _b16(dout) else lib().stem_bwd_weight_bn      # (bf16 storage: the gradient arrives as bf16, y is fp32)
    fn(x.data_ptr(), y.data_ptr(), dout.data_ptr(), lddo, mean.data_ptr(), invstd.data_ptr(), _p(gamma), _p(beta),
       act, float(slope), sums.data_ptr(), rows, n, h, wd, c, dw.data_ptr(), slabs.data_ptr(), _s())
    return dw, sums[1], sums[0]


# BatchNorm's num_batches_tracked counters of a forward: one kernel each when bumped where the layer runs (23 six-microsecond launches on
# the head's chain); a forward that calls batches_begin() / batches_end() bumps them with ONE _foreach_add_ at its end
_nbt = None


def bump_batches(bn) -> None:
    if _nbt is not None:
        _nbt.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


def batches_begin() -> None:
    global _nbt
    _nbt = []


def batches_end() -> None:
    global _nbt
    pend, _nbt = _nbt, None
    if pend:
        torch._foreach_add_(pend, 1)


LANGUAGE_LATE = True       # A/B switch (captured steps): the language branch starts behind the backbone's register-bank layers
LANGUAGE_BWD_HOPS = 1             # (model.finish_backward: queue steering in captured steps; fp32: 97.7 / 95.6 / 96.6 ms with 0 / 1 / 2)
LANGUAGE_BWD_HOPS_B16 = 0         # bf16 storage: the branch shares the weight gradients' queue (59.9 ms against 60.7 with a queue of its own)
LANGUAGE_BWD_DEFERRED = True      # A/B switch (graph.GraphedTrainStep): the language branch's backward as its own stage behind loss.backward()
STEM_FUSED_BWD = True      # A/B switch: False = bn_act_bwd (writes dy) + conv2d_bwd_weight for the stem
BN_TAP = True              # A/B switch: False = never ask a data gradient for the partial sums of the BatchNorm in front
BN_TAP_TRUNK = os.environ.get("DCN_BN_TAP_TRUNK", "1") != "0"
# ... also from the stride-1 layers (the partial-sum epilogues of csrc/conv1.hip / conv3.hip): 53 of the 87 reduce passes of a step go
# (-2.7 ms of kernel time) and the data gradients' epilogues take most of it back: -0.3 .. +0.05 ms on the step in four A/B runs
# (2: the 3x3 layers only, 3: the 1x1 layers only)


def act_bwd(out, dout, slope):
    c = out.shape[-1]
    rows = out.numel() // c
    dy = torch.empty_like(out)
    lib().act_bwd(out.data_ptr(), dout.data_ptr(), dout.stride(-2), float(slope), rows, c, dy.data_ptr(), _s())
    return dy


# ---- co-attention -----------------------------------------------------------------------------------
def coattn_fwd(f1, f2, out1, out2, temperature):
    """f1,f2 (b,hw,c) views with pixel stride ldf (last dim contiguous); out1/out2 (b,hw,c) views with
    stride ldo (out2 may be None).  Returns the saved (E, rinv, cinv)."""
    b, hw, c = f1.shape
    dev = f1.device
    E = torch.empty(lib().coattn_saved_size(b, hw, c), dtype=torch.float32, device=dev)
    rc = torch.empty((2, b, hw), dtype=torch.float32, device=dev)
    ws = scratch(lib().coattn_fwd_ws(b, hw, c), dev, slot=0)
    assert f1.stride() == f2.stride() and (out2 is None or out1.stride() == out2.stride())
    lib().coattn_fwd(f1.data_ptr(), f2.data_ptr(), f1.stride(1), f1.stride(0), out1.data_ptr(), _p(out2), out1.stride(1),
                     out1.stride(0), E.data_ptr(), rc[0].data_ptr(), rc[1].data_ptr(), ws.data_ptr(), b, hw, c,
                     float(temperature), _s())
    return E, rc


def coattn_bwd(f1, f2, d_out1, d_out2, out1, out2, E, rc, d_f1, d_f2, accumulate, temperature):
    b, hw, c = f1.shape
    ws = scratch(lib().coattn_bwd_ws(b, hw, c), f1.device, slot=0)
    assert f1.stride() == f2.stride() and d_out1.stride() == d_out2.stride() and out1.stride() == out2.stride()
    assert d_f1.stride() == d_f2.stride()
    lib().coattn_bwd(f1.data_ptr(), f2.data_ptr(), f1.stride(1), f1.stride(0),
                     d_out1.data_ptr(), d_out2.data_ptr(), d_out1.stride(1), d_out1.stride(0),
                     out1.data_ptr(), out2.data_ptr(), out1.stride(1), out1.stride(0),
                     E.data_ptr(), rc[0].data_ptr(), rc[1].data_ptr(),
                     d_f1.data_ptr(), d_f2.data_ptr(), d_f1.stride(1), d_f1.stride(0), int(accumulate), ws.data_ptr(),
                     b, hw, c, float(temperature), _s())


def gemm3_presplit(x, amax, out=None):
    """x (b, rows, c) fp32 view (last dim contiguous) -> its f16 two-piece split form (same shape / bytes; csrc/gemm3.hip), scaled by the
    power of two of the abs-max word ``amax``.  out=x splits in place."""
    b, rows, c = x.shape
    if out is None:
        out = torch.empty((b, rows, c), dtype=torch.float32, device=x.device)
    lib().gemm3_presplit(x.data_ptr(), x.stride(1), x.stride(0), out.data_ptr(), out.stride(1), out.stride(0), b, rows, c, amax.data_ptr(), _s())
    return out


def gemm3(a, b, out, m, n, k, amax_a, amax_b, a_t=False, b_t=False, row_scale=None, accumulate=False):
    """out[i] (+)= diag(row_scale[i]) op(a[i]) op(b[i])^T on operands in split form (gemm3_presplit).  a: (batch, m, >=k) or, a_t,
    (batch, k, >=m); b: (batch, n, >=k) or, b_t, (batch, k, >=n); out (batch, m, >=n) fp32; all last-dim contiguous views."""
    lib().gemm3(a.data_ptr(), a.stride(1), a.stride(0), int(a_t), b.data_ptr(), b.stride(1), b.stride(0), int(b_t),
                out.data_ptr(), out.stride(1), out.stride(0), _p(row_scale), 0 if row_scale is None else row_scale.stride(0),
                m, n, k, a.shape[0], int(accumulate), amax_a.data_ptr(), amax_b.data_ptr(), _s())
    return out


# ---- scoring ------------------------------------------------------------------------------------------
def l2norm_score_fwd(x, q=None, rows_per_image=0, out=None, want_flip=False, out_scale=1.0, accumulate=False):
    """x (...,c) rows (pixel stride = x.stride(-2)).  Returns (out, norm, score|None, score_flip|None); score_flip scores
    row r against q[N-1-img(r)] (the caller-side neg_sim_score, train_DCNet.py:623-627)."""
    c = x.shape[-1]
    rows = x.numel() // c
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _rows(x, "l2norm x"); _rows(out, "l2norm out")
    norm = torch.empty(rows, dtype=torch.float32, device=x.device)
    score = torch.empty(rows, dtype=torch.float32, device=x.device) if q is not None else None
    flip = torch.empty(rows, dtype=torch.float32, device=x.device) if (q is not None and want_flip) else None
    lib().l2norm_score_fwd(x.data_ptr(), x.stride(-2), out.data_ptr(), out.stride(-2), norm.data_ptr(), _p(q), _p(score), _p(flip),
                           rows, rows_per_image, c, float(out_scale), int(accumulate), _s())
    return out, norm, score, flip


def l2norm_score_bwd(out, norm, dout, q, dscore, rows_per_image, want_dq=True, dscore_flip=None):
    c = out.shape[-1]
    rows = out.numel() // c
    _rows(out, "l2norm_bwd out")
    if dout is not None:
        _rows(dout, "l2norm_bwd dout")
    dx = torch.empty(out.shape, dtype=torch.float32, device=out.device)
    dq = None
    if q is not None and (dscore is not None or dscore_flip is not None) and want_dq:
        dq = torch.empty((rows // rows_per_image, c), dtype=torch.float32, device=out.device)
    lib().l2norm_score_bwd(out.data_ptr(), out.stride(-2), norm.data_ptr(), _p(dout), 0 if dout is None else dout.stride(-2),
                           _p(q), _p(dscore), _p(dscore_flip), dx.data_ptr(), c, _p(dq), rows, rows_per_image, c, _s())
    return dx, dq


def rowdot_fwd(x, q, rows_per_image, flip=False):
    """score[row] = <x[row], q[img(row)]> (flip: q[N-1-img]) on rows that are not normalised here."""
    c = x.shape[-1]
    rows = x.numel() // c
    _rows(x, "rowdot x"); _chk(q, "rowdot q")
    score = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib().rowdot_fwd(x.data_ptr(), x.stride(-2), q.data_ptr(), int(flip), score.data_ptr(), rows, rows_per_image, c, _s())
    return score


def rowdot_bwd(x, q, dscore, rows_per_image, flip=False, want_dx=True, want_dq=True):
    c = x.shape[-1]
    rows = x.numel() // c
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_dx else None
    dq = torch.empty((rows // rows_per_image, c), dtype=torch.float32, device=x.device) if want_dq else None
    lib().rowdot_bwd(x.data_ptr(), x.stride(-2), q.data_ptr(), int(flip), dscore.data_ptr(), _p(dx), c, _p(dq), rows, rows_per_image, c, _s())
    return dx, dq


# ---- phrase attention ----------------------------------------------------------------------------------
def phrase_attn_fwd(context, embedded, ids, w0, b0, w1=None, b1=None, normalize=True):
    """Returns (attn (H,N,L), out (H,N,E), vnorm (H,N)|None) for H = 1 or 2 heads."""
    n, L, d = context.shape
    e = embedded.shape[2]
    _chk(context, "phrase context"); _chk(embedded, "phrase embedded")
    if not (ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous()):
        raise ValueError("phrase_attn: ids must be a contiguous int64 CUDA tensor")
    heads = 1 if w1 is None else 2
    attn = torch.empty((heads, n, L), dtype=torch.float32, device=context.device)
    out = torch.empty((heads, n, e), dtype=torch.float32, device=context.device)
    vnorm = torch.empty((heads, n), dtype=torch.float32, device=context.device) if normalize else None
    lib().phrase_attn_fwd(context.data_ptr(), embedded.data_ptr(), ids.data_ptr(), w0.data_ptr(), b0.data_ptr(), _p(w1), _p(b1),
                          attn.data_ptr(), out.data_ptr(), _p(vnorm), n, L, d, e, int(normalize), _s())
    return attn, out, vnorm


def phrase_attn_bwd(context, embedded, w0, w1, attn, out, vnorm, dout, normalize=True):
    """Returns (dcontext, dembedded, dwb) with dwb = [dw_0 (d) | dw_1 (d) | db_0 | db_1] (one head: [dw_0 | db_0])."""
    n, L, d = context.shape
    e = embedded.shape[2]
    heads = 1 if w1 is None else 2
    dctx = torch.empty_like(context); demb = torch.empty_like(embedded)
    dwb = torch.empty(heads * d + heads, dtype=torch.float32, device=context.device)
    ws = scratch(lib().phrase_attn_bwd_ws(n, d, heads), context.device, slot=0)
    _chk(dout, "phrase dout")
    lib().phrase_attn_bwd(context.data_ptr(), embedded.data_ptr(), w0.data_ptr(), _p(w1), attn.data_ptr(), out.data_ptr(), _p(vnorm),
                          dout.data_ptr(), dctx.data_ptr(), demb.data_ptr(), dwb.data_ptr(), ws.data_ptr(), n, L, d, e, int(normalize), _s())
    return dctx, demb, dwb


def colsum(x2d):
    """out[c] = sum_r x2d[r, c] (deterministic, for small row counts); rows may be strided."""
    if not (x2d.is_cuda and x2d.dtype == torch.float32 and x2d.dim() == 2 and x2d.stride(1) == 1):
        raise ValueError("colsum: expected an fp32 CUDA [rows][cols] view with contiguous columns")
    rows, cols = x2d.shape
    out = torch.empty(cols, dtype=torch.float32, device=x2d.device)
    lib().colsum(x2d.data_ptr(), x2d.stride(0), rows, cols, out.data_ptr(), _s())
    return out


# ---- fusion layer constants, small language-branch ops (csrc/fusion.hip) ------------------------------------------------
def fusion_prefill(a_img, coord2d, w3):
    """out (n,hw,co) = a_img[n,None,:] + coord2d (hw,8) @ w3 (co,8 view, row-strided)^T."""
    n, co = a_img.shape
    hw = coord2d.shape[0]
    _chk(a_img, "fusion_prefill A"); _chk(coord2d, "fusion_prefill coord")
    assert w3.shape == (co, 8) and w3.stride(1) == 1
    out = torch.empty((n, hw, co), dtype=torch.float32, device=a_img.device)
    lib().fusion_prefill(a_img.data_ptr(), coord2d.data_ptr(), w3.data_ptr(), w3.stride(0), out.data_ptr(), n, hw, co, _s())
    return out


def fusion_bwd(dy, coord2d, flang, dweight, e):
    """dy (n,h,w,co) contiguous.  Writes dW2 / dW3 into columns [e, 2e) / [2e, 2e+8) of ``dweight`` (co, 2e+8) and returns d_img (n,co)."""
    n, co = dy.shape[0], dy.shape[-1]
    hw = dy.numel() // (n * co)
    _chk(dy, "fusion_bwd dy"); _chk(flang, "fusion_bwd flang"); _chk(dweight, "fusion_bwd dweight")
    d_img = torch.empty((n, co), dtype=torch.float32, device=dy.device)
    ws = scratch(lib().fusion_bwd_ws(n, co), dy.device, slot=0)
    ld = dweight.stride(0)
    lib().fusion_bwd(dy.data_ptr(), coord2d.data_ptr(), flang.data_ptr(), ws.data_ptr(), d_img.data_ptr(),
                     dweight[:, e:].data_ptr(), dweight[:, 2 * e:].data_ptr(), ld, n, hw, co, e, _s())
    return d_img


def _ids(ids: torch.Tensor, name: str) -> torch.Tensor:
    """Token ids as the kernels read them: contiguous int64 on the device.  The reference's `(ids != 0).sum(1)` and nn.Embedding
    take int32 ids as well (DCNet_model.py:150,168), so integer dtypes are widened; anything else raises."""
    if not ids.is_cuda:
        raise ValueError(f"{name}: token ids must be a CUDA tensor (no CPU path), got {ids.device}")
    if ids.dtype not in (torch.int64, torch.int32, torch.int16, torch.uint8, torch.int8):
        raise ValueError(f"{name}: token ids must be an integer tensor, got {ids.dtype}")
    return ids.to(torch.int64).contiguous()


def row_lengths(ids):
    ids = _ids(ids, "row_lengths")
    n, L = ids.shape
    out = torch.empty(n, dtype=torch.int64, device=ids.device)
    lib().row_lengths(ids.data_ptr(), n, L, out.data_ptr(), _s())
    return out


def embedding_fwd(ids, table):
    ids = _ids(ids, "embedding_fwd"); _chk(table, "embedding_fwd table")
    if ids.device != table.device:
        raise ValueError("embedding_fwd: ids and table on different devices")
    v, e = table.shape
    out = torch.empty(tuple(ids.shape) + (e,), dtype=torch.float32, device=table.device)
    lib().embedding_fwd(ids.data_ptr(), table.data_ptr(), out.data_ptr(), ids.numel(), e, v, _s())
    return out


def embedding_bwd(ids, dout, vocab):
    ids = _ids(ids, "embedding_bwd"); _chk(dout, "embedding_bwd dout")
    if ids.device != dout.device or dout.numel() != ids.numel() * dout.shape[-1]:
        raise ValueError("embedding_bwd: dout must be (ids.shape, e) on the ids' device")
    e = dout.shape[-1]
    dt = torch.empty((vocab, e), dtype=torch.float32, device=dout.device)
    lib().embedding_bwd(ids.data_ptr(), dout.data_ptr(), dt.data_ptr(), ids.numel(), e, vocab, _s())
    return dt


# ---- cross-scale head tail -------------------------------------------------------------------------------
import ctypes as _ct


def _ptrs(ts):
    """Host array of three device pointers (None -> NULL); keeps nothing alive: callers hold the tensors."""
    return (_ct.c_void_p * 3)(*[(None if t is None else t.data_ptr()) for t in ts])


def _ints(vs):
    return (_ct.c_int * 3)(*[int(v) for v in vs])


def locemb_fwd(coord, w, b, gamma, beta, bn, training, count):
    P = coord.shape[0]
    dev = coord.device
    e8 = torch.empty((P, 8), dtype=torch.float32, device=dev); xh = torch.empty((P, 8), dtype=torch.float32, device=dev)
    stat = torch.empty(16, dtype=torch.float32, device=dev); mom = torch.empty(72, dtype=torch.float64, device=dev)
    lib().locemb_fwd(coord.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bn.running_mean.data_ptr(),
                     bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps), int(training), int(count), P,
                     e8.data_ptr(), xh.data_ptr(), stat.data_ptr(), mom.data_ptr(), _s())
    return e8, xh, stat, mom


def locemb_bwd(coord, w, gamma, beta, xh, stat, de_a, de_b, dmom, training):
    grads = torch.empty(88, dtype=torch.float32, device=coord.device)
    lib().locemb_bwd(coord.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), xh.data_ptr(), stat.data_ptr(),
                     _p(de_a), _p(de_b), _p(dmom), int(training), coord.shape[0], grads.data_ptr(), _s())
    return grads


def head_obj(logits, sims, e8):
    """logits[s] (B,H,W,ld) NHWC, sims[s] (B,HW).  Returns (only_obj[3] (B,h,w), obj_map (B,P), objn (B), X (B*8,Ppad))."""
    B = logits[0].shape[0]
    dev = e8.device
    hw = [t.shape[1] * t.shape[2] for t in logits]
    P = sum(hw); Ppad = pad32(P)
    only = [torch.empty((B, t.shape[1], t.shape[2]), dtype=torch.float32, device=dev) for t in logits]
    obj_map = torch.empty((B, P), dtype=torch.float32, device=dev); objn = torch.empty(B, dtype=torch.float32, device=dev)
    X = torch.empty((B * 8, Ppad), dtype=torch.float32, device=dev)
    for t in logits:
        _chk(t, "head_obj logits")
    lib().head_obj(_ptrs(logits), _ints([t.shape[3] for t in logits]), _ptrs(sims), _ints(hw), e8.data_ptr(), _ptrs(only),
                   obj_map.data_ptr(), objn.data_ptr(), X.data_ptr(), B, _s())
    return only, obj_map, objn, X


def pad_rows(src, cols_out):
    """(rows, cols) contiguous -> (rows, cols_out) with zero-filled / dropped tail columns."""
    rows, cols = src.shape
    _chk(src, "pad_rows")
    dst = torch.empty((rows, cols_out), dtype=torch.float32, device=src.device)
    lib().pad_rows(src.data_ptr(), cols, dst.data_ptr(), cols_out, rows, min(cols, cols_out), cols_out, _s())
    return dst


def locbn_fwd(M, mom, bias, gamma, beta, bn, training, count):
    B = M.shape[0]
    Mp = torch.empty_like(M); bp = torch.empty(512, dtype=torch.float32, device=M.device)
    saved = torch.empty((3, 512), dtype=torch.float32, device=M.device)
    lib().locbn_fwd(M.data_ptr(), mom.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bn.running_mean.data_ptr(),
                    bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps), int(training), B, int(count), M.shape[2],
                    Mp.data_ptr(), bp.data_ptr(), saved.data_ptr(), _s())
    return Mp, bp, saved


def locbn_bwd(M, mom, bias, gamma, running_mean, saved, dMp, dbp, training, count):
    """dMp (B,8,512): a view whose image stride may exceed 8*512 (rows 0-7 of dcn_locmod_bwd's dsum)."""
    B = M.shape[0]
    assert dMp.stride(2) == 1 and dMp.stride(1) == M.shape[2]
    dM = torch.empty_like(M); out = torch.empty((3, 512), dtype=torch.float32, device=M.device)
    dmom = torch.empty(72, dtype=torch.float64, device=M.device) if training else None
    lib().locbn_bwd(M.data_ptr(), mom.data_ptr(), bias.data_ptr(), gamma.data_ptr(), running_mean.data_ptr(), saved.data_ptr(),
                    dMp.data_ptr(), dMp.stride(0), dbp.data_ptr(), int(training), B, int(count), M.shape[2], dM.data_ptr(), out.data_ptr(),
                    _p(dmom), _s())
    return dM, out, dmom


def head_final_fwd(logits, sims, loc_map):
    B = logits[0].shape[0]
    dev = loc_map.device
    hw = [t.shape[1] * t.shape[2] for t in logits]
    outbox = [torch.empty((B, 15, t.shape[1], t.shape[2]), dtype=torch.float32, device=dev) for t in logits]
    loc = [torch.empty((B, t.shape[1], t.shape[2]), dtype=torch.float32, device=dev) for t in logits]
    mm = torch.empty((B, 4), dtype=torch.float32, device=dev)
    lib().head_final_fwd(_ptrs(logits), _ints([t.shape[3] for t in logits]), _ptrs(sims), _ints(hw), loc_map.data_ptr(),
                         _ptrs(outbox), _ptrs(loc), mm.data_ptr(), B, _s())
    return outbox, loc, mm


def head_dloc(logits, sims, loc, d_outbox, d_loc, mm):
    B = logits[0].shape[0]
    hw = [t.shape[1] * t.shape[2] for t in logits]
    dloc_map = torch.empty((B, sum(hw)), dtype=torch.float32, device=mm.device)
    lib().head_dloc(_ptrs(logits), _ints([t.shape[3] for t in logits]), _ptrs(sims), _ints(hw), _ptrs(loc), _ptrs(d_outbox),
                    _ptrs(d_loc), mm.data_ptr(), dloc_map.data_ptr(), B, _s())
    return dloc_map


def head_fold(dX, e8, obj_map):
    B, P = obj_map.shape
    dobj = torch.empty_like(obj_map); de8 = torch.empty_like(e8)
    lib().head_fold(dX.data_ptr(), e8.data_ptr(), obj_map.data_ptr(), B, P, dX.shape[1], dobj.data_ptr(), de8.data_ptr(), _s())
    return dobj, de8


def head_dlogits(logits, sims, loc, only, d_outbox, d_only, obj_map, objn, dobj_map):
    B = logits[0].shape[0]
    hw = [t.shape[1] * t.shape[2] for t in logits]
    dlogits = [torch.empty_like(t) for t in logits]
    dsim = [torch.empty_like(t) for t in sims]
    lib().head_dlogits(_ptrs(logits), _ints([t.shape[3] for t in logits]), _ptrs(sims), _ints(hw), _ptrs(loc), _ptrs(only),
                       _ptrs(d_outbox), _ptrs(d_only), obj_map.data_ptr(), objn.data_ptr(), _p(dobj_map), _ptrs(dlogits), _ptrs(dsim), B, _s())
    return dlogits, dsim


# ---- correspondence sampling ---------------------------------------------------------------------------------
def gemm_nt_batched(a, b, out=None):
    """a (B,M,K), b (B,N,K) (last dim contiguous, uniform strides) -> (B,M,N) = a . b^T per batch item."""
    B, M, K = a.shape
    N = b.shape[1]
    if out is None:
        out = torch.empty((B, M, N), dtype=torch.float32, device=a.device)
    lib().gemm_nt_batched(a.data_ptr(), a.stride(1), a.stride(0), b.data_ptr(), b.stride(1), b.stride(0), out.data_ptr(), out.stride(1),
                          out.stride(0), M, N, K, B, _s())
    return out


def k9_fwd(fv, raw_neg, top_k):
    """fv (N,HW,E) contiguous; raw_neg (N/2,top_k,neg_n) int64.  Returns (index, neg_idx, frame, corr, negf)."""
    n, hw, e = fv.shape
    b = n // 2
    neg_n = raw_neg.shape[2]
    dev = fv.device
    pairs = fv.view(b, 2, hw, e)
    cmap = gemm_nt_batched(pairs[:, 0], pairs[:, 1])                               # (b,hw,hw): [i*hw + j]   :390
    index = torch.empty((b, top_k), dtype=torch.int64, device=dev); neg_idx = torch.empty((b, top_k, neg_n), dtype=torch.int64, device=dev)
    frame = torch.empty((b, top_k, e), dtype=torch.float32, device=dev); corr = torch.empty_like(frame)
    negf = torch.empty((b, top_k, neg_n, e), dtype=torch.float32, device=dev)
    lib().k9_fwd(cmap.data_ptr(), fv.data_ptr(), raw_neg.data_ptr(), b, hw, e, top_k, neg_n, index.data_ptr(), neg_idx.data_ptr(),
                 frame.data_ptr(), corr.data_ptr(), negf.data_ptr(), _s())
    return index, neg_idx, frame, corr, negf


def k9_bwd(index, neg_idx, d_frame, d_corr, d_neg, hw):
    b, top_k, neg_n = neg_idx.shape
    e = d_frame.shape[2]
    dfv = torch.empty((2 * b, hw, e), dtype=torch.float32, device=d_frame.device)
    lib().k9_bwd(index.data_ptr(), neg_idx.data_ptr(), d_frame.data_ptr(), d_corr.data_ptr(), d_neg.data_ptr(), b, hw, e, top_k, neg_n,
                 dfv.data_ptr(), _s())
    return dfv


def colnorm_fwd(v):
    n, hw, e = v.shape
    vit = torch.empty_like(v); cn = torch.empty((n, e), dtype=torch.float32, device=v.device)
    lib().colnorm_fwd(v.data_ptr(), n, hw, e, vit.data_ptr(), cn.data_ptr(), _s())
    return vit, cn


def colnorm_bwd(vit, cn, dq, extra, n_extra):
    n, hw, e = vit.shape
    dv = torch.empty_like(vit)
    lib().colnorm_bwd(vit.data_ptr(), cn.data_ptr(), dq.data_ptr(), _p(extra), n_extra, n, hw, e, dv.data_ptr(), _s())
    return dv


def lagnorm_fwd(context):
    n, L, d = context.shape
    lag = torch.empty((n, L, d // 2), dtype=torch.float32, device=context.device)
    ln = torch.empty((n, d // 2), dtype=torch.float32, device=context.device)
    lib().lagnorm_fwd(context.data_ptr(), n, L, d, lag.data_ptr(), ln.data_ptr(), _s())
    return lag, ln


def crossmap(lag, vit, conv_w, conv_b, want_map=False):
    n, L, e = lag.shape
    hw = vit.shape[1]
    cols = torch.empty((n, hw), dtype=torch.int64, device=lag.device)
    lv = torch.empty((n, L, hw), dtype=torch.float32, device=lag.device) if want_map else None
    lib().crossmap(lag.data_ptr(), vit.data_ptr(), conv_w.data_ptr(), conv_b.data_ptr(), n, L, hw, e, cols.data_ptr(), _p(lv), _s())
    return cols, lv


def k14_gather(lag, vit, cols, neg):
    n, L, e = lag.shape
    hw = vit.shape[1]
    neg_n = neg.shape[2]
    lag_pos = torch.empty((n, hw, 1, e), dtype=torch.float32, device=lag.device)
    neg_cross = torch.empty((n, hw, neg_n, e), dtype=torch.float32, device=lag.device)
    lib().k14_gather(lag.data_ptr(), vit.data_ptr(), cols.data_ptr(), neg.data_ptr(), n, L, hw, e, neg_n, lag_pos.data_ptr(),
                     neg_cross.data_ptr(), _s())
    return lag_pos, neg_cross


def k14_negscatter(d_neg, csr_off, csr_src, hw):
    e = d_neg.shape[-1]
    extra = torch.empty((hw, e), dtype=torch.float32, device=d_neg.device)
    lib().k14_negscatter(d_neg.data_ptr(), csr_off.data_ptr(), csr_src.data_ptr(), hw, e, extra.data_ptr(), _s())
    return extra


def k14_dlag(lag, ln, cols, d_k):
    n, L, e = lag.shape
    hw = cols.shape[1]
    dctx = torch.empty((n, L, 2 * e), dtype=torch.float32, device=lag.device)
    lib().k14_dlag(lag.data_ptr(), ln.data_ptr(), cols.data_ptr(), d_k.data_ptr(), n, L, hw, e, dctx.data_ptr(), _s())
    return dctx


# ---- losses, targets, decode --------------------------------------------------------------------------------------
def build_target(bbox, anchors, size):
    n = bbox.shape[0]
    ti = torch.empty((n, 4), dtype=torch.int32, device=bbox.device); tf = torch.empty((n, 4), dtype=torch.float32, device=bbox.device)
    _chk(bbox, "build_target bbox")
    lib().build_target(bbox.data_ptr(), anchors.data_ptr(), size, n, ti.data_ptr(), tf.data_ptr(), _s())
    return ti, tf


def target_dense(ti, tf, size):
    n = ti.shape[0]
    grids = [size // 32, size // 16, size // 8]
    box = [torch.zeros((n, 3, 5, g, g), dtype=torch.float32, device=ti.device) for g in grids]
    ctr = [torch.zeros((n, 5, g, g), dtype=torch.float32, device=ti.device) for g in grids]
    lib().target_dense(ti.data_ptr(), tf.data_ptr(), size, n, _ptrs(box), _ptrs(ctr), _s())
    return box, ctr


def dense_loss_fwd(outbox, sim, negsim, loc, ti, tf, size):
    n = ti.shape[0]
    dev = ti.device
    vals = torch.empty((n, 8), dtype=torch.float32, device=dev); lse = torch.empty((n, 2), dtype=torch.float32, device=dev)
    out = torch.empty(3, dtype=torch.float32, device=dev)
    for t in list(outbox) + list(sim) + list(negsim) + list(loc):
        _chk(t, "dense_loss map")
    lib().dense_loss_fwd(_ptrs(outbox), _ptrs(sim), _ptrs(negsim), _ptrs(loc), ti.data_ptr(), tf.data_ptr(), size, n, vals.data_ptr(),
                         lse.data_ptr(), out.data_ptr(), _s())
    return out, lse


def dense_loss_bwd(outbox, sim, negsim, loc, ti, tf, lse, gout, size):
    n = ti.shape[0]
    d_ob = [torch.empty_like(t) for t in outbox]; d_sim = [torch.empty_like(t) for t in sim]
    d_ns = [torch.empty_like(t) for t in negsim]; d_loc = [torch.empty_like(t) for t in loc]
    _chk(gout, "dense_loss gout")
    lib().dense_loss_bwd(_ptrs(outbox), _ptrs(sim), _ptrs(negsim), _ptrs(loc), ti.data_ptr(), tf.data_ptr(), lse.data_ptr(), gout.data_ptr(),
                         size, n, _ptrs(d_ob), _ptrs(d_sim), _ptrs(d_ns), _ptrs(d_loc), _s())
    return d_ob, d_sim, d_ns, d_loc


def contrastive_fwd(q, pos, neg, temperature):
    """q, pos [rows][e], neg [rows][m][e] contiguous.  Returns the mean InfoNCE loss (0-dim tensor)."""
    e = q.shape[-1]
    rows = q.numel() // e
    m = neg.numel() // (rows * e)
    for t in (q, pos, neg):
        _chk(t, "contrastive")
    rl = scratch(rows, q.device, slot=4)
    loss = torch.empty((), dtype=torch.float32, device=q.device)
    lib().contrastive_fwd(q.data_ptr(), pos.data_ptr(), neg.data_ptr(), rows, e, m, float(temperature), rl.data_ptr(), loss.data_ptr(), _s())
    return loss


def contrastive_bwd(q, pos, neg, temperature, gout):
    e = q.shape[-1]
    rows = q.numel() // e
    m = neg.numel() // (rows * e)
    dq = torch.empty_like(q); dpos = torch.empty_like(pos); dneg = torch.empty_like(neg)
    lib().contrastive_bwd(q.data_ptr(), pos.data_ptr(), neg.data_ptr(), rows, e, m, float(temperature), gout.data_ptr(),
                          dq.data_ptr(), dpos.data_ptr(), dneg.data_ptr(), _s())
    return dq, dpos, dneg


def decode_boxes(outbox, anchors, size, want_cells=False):
    n = outbox[0].shape[0]
    for t in outbox:
        _chk(t, "decode outbox")
    boxes = torch.empty((n, 4), dtype=torch.float32, device=outbox[0].device)
    cells = torch.empty((n, 3), dtype=torch.int32, device=outbox[0].device) if want_cells else None
    lib().decode_boxes(_ptrs(outbox), anchors.data_ptr(), size, n, boxes.data_ptr(), _p(cells), _s())
    return (boxes, cells) if want_cells else boxes


def post_topk(outbox, corr_feat, anchors, size, topk, ratio, dw, dh, frame_hw):
    """csrc/post.hip: outbox[s] (B,15,g,g) contiguous fp32, corr_feat[s] (B,E,g,g) fp32 with any strides, ratio/dw/dh (B,) fp32,
    frame_hw (B,2) int64.  Returns boxes (B,k,4), scores (B,k), feats (B,k,E), cells (B,k,4) int64."""
    B = outbox[0].shape[0]
    dev = outbox[0].device
    E = corr_feat[0].shape[1]
    for t in outbox:
        _chk(t, "post_topk outbox")
    for t in corr_feat:
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.shape[0] == B and t.shape[1] == E):
            raise ValueError("post_topk: corr_feat[s] must be an fp32 CUDA tensor (B,E,g,g)")
    for t, nm in ((ratio, "ratio"), (dw, "dw"), (dh, "dh")):
        _chk(t, "post_topk " + nm)
        if t.numel() != B:
            raise ValueError(f"post_topk: {nm} must have one entry per clip")
    if not (frame_hw.is_cuda and frame_hw.dtype == torch.int64 and frame_hw.is_contiguous() and tuple(frame_hw.shape) == (B, 2)):
        raise ValueError("post_topk: frame_hw must be a contiguous int64 CUDA tensor (B,2)")
    grids = [int(o.shape[-1]) for o in outbox]
    strides = (_ct.c_int64 * 12)(*[int(v) for t in corr_feat for v in t.stride()])
    boxes = torch.empty((B, topk, 4), dtype=torch.float32, device=dev); scores = torch.empty((B, topk), dtype=torch.float32, device=dev)
    feats = torch.empty((B, topk, E), dtype=torch.float32, device=dev); cells = torch.empty((B, topk, 4), dtype=torch.int64, device=dev)
    lib().post_topk(_ptrs(outbox), _ptrs(corr_feat), strides, _ints(grids), anchors.data_ptr(), size, B, E, topk, ratio.data_ptr(),
                    dw.data_ptr(), dh.data_ptr(), frame_hw.data_ptr(), boxes.data_ptr(), scores.data_ptr(), feats.data_ptr(),
                    cells.data_ptr(), _s())
    return boxes, scores, feats, cells


def post_fusion(center, ref, ref_score, valid=None):
    """csrc/post.hip: center (B,k,E), ref (B,R,k,E), ref_score (B,R,k) fp32 contiguous; valid (B,R) bool/uint8 or None.
    Returns (best (B,) int64, fused (B,k))."""
    _chk(center, "post_fusion center"); _chk(ref, "post_fusion ref"); _chk(ref_score, "post_fusion ref_score")
    B, k, E = center.shape
    R = ref.shape[1]
    if tuple(ref.shape) != (B, R, k, E) or tuple(ref_score.shape) != (B, R, k):
        raise ValueError("post_fusion: shapes must be center (B,k,E), ref (B,R,k,E), ref_score (B,R,k)")
    v = None
    if valid is not None:
        if not valid.is_cuda or tuple(valid.shape) != (B, R):
            raise ValueError("post_fusion: valid must be a CUDA tensor (B,R)")
        v = valid.to(torch.uint8).contiguous()
    fused = torch.empty((B, k), dtype=torch.float32, device=center.device); best = torch.empty(B, dtype=torch.int64, device=center.device)
    lib().post_fusion(center.data_ptr(), ref.data_ptr(), ref_score.data_ptr(), _p(v), B, k, R, E, fused.data_ptr(), best.data_ptr(), _s())
    return best, fused


def box_iou(b1, b2):
    _chk(b1, "box_iou"); _chk(b2, "box_iou")
    iou = torch.empty(b1.shape[0], dtype=torch.float32, device=b1.device)
    lib().box_iou(b1.data_ptr(), b2.data_ptr(), b1.shape[0], iou.data_ptr(), _s())
    return iou


# ---- data movers ---------------------------------------------------------------------------------------
def upsample2_into(src, dst_view):
    """src (N,h,w,c) -> dst_view (N,2h,2w,c) (a channel slice of a wider NHWC buffer)."""
    n, h, w, c = src.shape
    if _b16(src):
        lib().upsample2_nhwc_b16(src.data_ptr(), src.stride(2), dst_view.data_ptr(), dst_view.stride(2), n, h, w, c, _s())
        return
    lib().upsample2_nhwc(src.data_ptr(), src.stride(2), dst_view.data_ptr(), dst_view.stride(2), n, h, w, c, _s())


def upsample2_bwd(ddst_view, dsrc, accumulate):
    n, h, w, c = dsrc.shape
    if _b16(dsrc):
        lib().upsample2_nhwc_bwd_b16(ddst_view.data_ptr(), ddst_view.stride(2), dsrc.data_ptr(), dsrc.stride(2), n, h, w, c,
                                     int(accumulate), _s())
        return
    lib().upsample2_nhwc_bwd(ddst_view.data_ptr(), ddst_view.stride(2), dsrc.data_ptr(), dsrc.stride(2), n, h, w, c,
                             int(accumulate), _s())


def copy_slice(src_view, dst_view, accumulate=False):
    if _b16(src_view) or _b16(dst_view):
        cast_rows(src_view, dst_view, accumulate)
        return
    c = src_view.shape[-1]
    rows = src_view.numel() // c
    _rows(src_view, "copy_slice src"); _rows(dst_view, "copy_slice dst")
    lib().copy_slice(src_view.data_ptr(), src_view.stride(-2), dst_view.data_ptr(), dst_view.stride(-2), rows, c,
                     int(accumulate), _s())


# ---- plain GEMMs / LSTM cell (language branch) ------------------------------------------------------------
def gemm_nt(a, b, bias=None, act=ACT_NONE, residual=None, out=None, accumulate=False):
    """out[M,N] (+)= act(a[M,K] @ b[N,K]^T + bias) + residual; a/residual/out may be row-strided views."""
    m, k = a.shape
    n = b.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    lib().gemm_nt(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), m, n, k,
                  _p(bias), act, _p(residual), 0 if residual is None else residual.stride(0), int(accumulate), _s())
    return out


def gemm_nn(a, b, out=None, accumulate=False, kvalid=0):
    """out[M,N] (+)= a[M,K] @ b[K,N]."""
    m, k = a.shape
    n = b.shape[1]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    lib().gemm_nn(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), m, n, k, kvalid,
                  int(accumulate), _s())
    return out


def gemm_tn(a, b, out=None, accumulate=False):
    """out[M,N] (+)= a[K,M]^T @ b[K,N]."""
    k, m = a.shape
    n = b.shape[1]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    lib().gemm_tn(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), m, n, k,
                  int(accumulate), _s())
    return out


def lstm_cell_fwd(gates, c_prev, h_prev, lens, t, act, c_out, h_out, y):
    n, h4 = gates.shape
    lib().lstm_cell_fwd(gates.data_ptr(), c_prev.data_ptr(), h_prev.data_ptr(), _p(lens), t, act.data_ptr(), c_out.data_ptr(),
                        h_out.data_ptr(), y.data_ptr(), y.stride(0), n, h4 // 4, _s())


def lstm_cell_bwd(dy, dh_rec, dc_next, act, c_prev, lens, t, dgates, dc_prev, dh_pass):
    n, h = c_prev.shape
    lib().lstm_cell_bwd(dy.data_ptr(), dy.stride(0), _p(dh_rec), _p(dc_next), act.data_ptr(), c_prev.data_ptr(), _p(lens), t,
                        dgates.data_ptr(), dgates.stride(0), dc_prev.data_ptr(), dh_pass.data_ptr(), n, h, _s())


_lstm_sync = {}


def _bilstm_sync(device):
    key = (torch.device(device).index, torch.cuda.current_stream().cuda_stream)
    t = _lstm_sync.get(key)
    if t is None:
        t = torch.zeros(int(lib().bilstm_sync_bytes()) // 4, dtype=torch.int32, device=device)
        _lstm_sync[key] = t
    return t


def bilstm_fwd(xg, whh_f, whh_r, bhh_f, bhh_r, lens):
    """xg (2,N,L,4H).  Returns (out (N,L,2H), hprev, cprev (2,N,L,H), acts (2,N,L,5H))."""
    _, n, L, h4 = xg.shape
    H = h4 // 4
    dev = xg.device
    out = torch.empty((n, L, 2 * H), dtype=torch.float32, device=dev)
    hprev = torch.zeros((2, n, L, H), dtype=torch.float32, device=dev)
    cprev = torch.empty((2, n, L, H), dtype=torch.float32, device=dev)
    acts = torch.empty((2, n, L, 5 * H), dtype=torch.float32, device=dev)
    for t in (xg, whh_f, whh_r, bhh_f, bhh_r):
        _chk(t, "bilstm_fwd")
    lib().bilstm_fwd(xg.data_ptr(), whh_f.data_ptr(), whh_r.data_ptr(), bhh_f.data_ptr(), bhh_r.data_ptr(), _p(lens), out.data_ptr(),
                     hprev.data_ptr(), cprev.data_ptr(), acts.data_ptr(), _bilstm_sync(dev).data_ptr(), n, L, H, _s())
    return out, hprev, cprev, acts


def bilstm_bwd(dout, whh_f, whh_r, acts, cprev, lens):
    _, n, L, H = cprev.shape
    dxg = torch.empty((2, n, L, 4 * H), dtype=torch.float32, device=dout.device)
    _chk(dout, "bilstm_bwd dout")
    if STEP_ABL & 2:             # timing-only ablation: the persistent BiLSTM backward is not launched (dxg uninitialised)
        return dxg.zero_()
    lib().bilstm_bwd(dout.data_ptr(), whh_f.data_ptr(), whh_r.data_ptr(), acts.data_ptr(), cprev.data_ptr(), _p(lens), dxg.data_ptr(),
                     _bilstm_sync(dout.device).data_ptr(), n, L, H, _s())
    return dxg


def bilstm_sync_error(device) -> bool:
    """True if a bounded spin of ANY persistent BiLSTM launch on this device — on whichever stream it ran (the model launches
    it on its language side stream) — gave up since that stream's buffer was created (the word is sticky: launches reset only
    their counters; a captured step re-creates its buffer on every replay, so there it speaks about the last replay).
    Host-synchronising (the whole device): bench.py, train.evaluate and train.save_checkpoint call it, so a hand-off that
    timed out cannot pass as a result."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    bufs = [t for (d, _s), t in _lstm_sync.items() if d == idx]
    if not bufs:
        return False
    torch.cuda.synchronize(idx)
    return any(bool(t[8].item()) for t in bufs)


def check_bilstm(device) -> None:
    if bilstm_sync_error(device):
        raise DcnError("persistent BiLSTM kernel: a grid hand-off spin timed out (the recurrence of some step is incomplete)")


def colsum_rows(x2d):
    """Column sums over MANY rows (alias of rows_sum for [rows][c] views with c % 4 == 0)."""
    return rows_sum(x2d.contiguous())


def rows_sum(x2d):
    """out[c] = sum over MANY rows of x2d[r, c] (two deterministic passes: per-block partials, then their sum)."""
    rows, c = x2d.shape
    _chk(x2d, "rows_sum")
    part = channel_stats(x2d)
    sums = torch.empty((2, c), dtype=torch.float32, device=x2d.device)
    ws = scratch(lib().bn_ws(c), x2d.device, slot=2)
    lib().bn_bwd_sums(part.data_ptr(), part.shape[0], c, sums.data_ptr(), ws.data_ptr(), _s())
    return sums[0]


# ---- location module core ------------------------------------------------------------------------------------
def locmod_fwd(E, Mp, bp, q):
    n, _, c = Mp.shape
    p = E.shape[0]
    loc = torch.empty((n, p), dtype=torch.float32, device=E.device)
    lib().locmod_fwd(E.data_ptr(), Mp.data_ptr(), bp.data_ptr(), q.data_ptr(), loc.data_ptr(), n, p, c, _s())
    return loc


def locmod_bwd(E, Mp, bp, q, dloc):
    n, _, c = Mp.shape
    p = E.shape[0]
    dE_part = torch.empty((n, p, 8), dtype=torch.float32, device=E.device)
    dsum = torch.empty((n, 10, c), dtype=torch.float32, device=E.device)
    ws = scratch(lib().locmod_bwd_ws(n, p), E.device, slot=0)
    lib().locmod_bwd(E.data_ptr(), Mp.data_ptr(), bp.data_ptr(), q.data_ptr(), dloc.data_ptr(), dE_part.data_ptr(),
                     dsum.data_ptr(), ws.data_ptr(), n, p, c, _s())
    return dE_part, dsum
