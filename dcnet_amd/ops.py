"""Thin tensor-level wrappers over the C ABI (dcnet_amd.lib): torch supplies device memory and
the stream, every computation below is a call into libdcnet_hip.so.

Layout convention inside the package: activations NHWC (contiguous torch tensors of shape
(N,H,W,C)), conv weights OHWI (Cout,k,k,Cin_padded).  Nothing here has a fallback path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .lib import DcnError, lib

ACT_NONE, ACT_LEAKY = 0, 1

# Matrix-pipe precision of the wide GEMM tiles (conv forward / data gradient / weight gradient, co-attention GEMMs):
#   "fp32"      fp32 accuracy on the bf16 matrix pipe (3 exact bf16 pieces per operand, 6 cross terms) — the default
#   "fp32_mfma" the native fp32 MFMA instruction on every tile
#   "bf16"      bf16 operands (round to nearest even), fp32 accumulate: BASELINE.json configs[2]; reduced precision,
#               builder-defined (the reference has no bf16 semantics); tensors in HBM stay fp32
#   "fp8"       forward and data-gradient tiles with OCP fp8 e4m3 operands (per-tensor power-of-two scales from an abs-max
#               pass, fp32 accumulate), weight gradient with bf16 operands: BASELINE.json configs[4]; reduced precision
PRECISIONS = {"fp32_mfma": 0, "fp32": 1, "bf16": 2, "fp8": 3}
_precision = "fp32"


def set_precision(mode: str) -> None:
    global _precision
    if mode not in PRECISIONS:
        raise ValueError(f"precision {mode!r}: expected one of {sorted(PRECISIONS)}")
    lib().set_tuning(b"precision", PRECISIONS[mode])
    _precision = mode


def get_precision() -> str:
    return _precision


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


def conv2d_fwd(x, w_ohwi, ksize, stride, scale=None, shift=None, act=ACT_NONE, slope=0.0,
               residual=None, out=None, want_stats=False, accumulate=False):
    """x (N,H,W,Cin) NHWC, w_ohwi (Cout,k,k,Cin) [or (Cout,64) for the stem].  Returns (y, stats)
    where stats is the [rows][2][Cout] partial-sum buffer (None unless want_stats)."""
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
    f8 = f8_scales(x, w_ohwi) if (_precision == "fp8" and cin != 4) else None
    lib().conv2d_fwd(x.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), n, h, wd, cin, cout, ksize, stride,
                     _p(scale), _p(shift), act, float(slope), _p(residual),
                     0 if residual is None else residual.stride(2), ldy, _p(stats), int(accumulate), _p(f8), _s())
    return out, stats


def conv2d_bwd_data(dy, w_ohwi, in_hw, ksize, stride, out=None, accumulate=False):
    """dy (N,Ho,Wo,Cout) (pixel stride may exceed Cout), w_ohwi (Cout,k,k,Cin) -> dx (N,H,W,Cin)."""
    n, ho, wo, cout = dy.shape
    cin = w_ohwi.shape[3]
    h, wd = in_hw
    if out is None:
        out = torch.empty((n, h, wd, cin), dtype=torch.float32, device=dy.device)
    wt = scratch(w_ohwi.numel(), dy.device, slot=1)
    f8 = f8_scales(dy, w_ohwi) if _precision == "fp8" else None
    lib().conv2d_bwd_data(dy.data_ptr(), dy.stride(2), w_ohwi.data_ptr(), wt.data_ptr(), out.data_ptr(),
                          n, h, wd, cin, cout, ksize, stride, int(accumulate), _p(f8), _s())
    return out


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


def conv2d_bwd_weight(x, dy, ksize, stride, cout=None, slot: int = 0):
    """x (N,H,W,Cin) NHWC (Cin == 4: stem), dy (N,Ho,Wo,Cout) -> dw OHWI (Cout,k,k,Cin) [(Cout,64) stem].
    ``slot`` selects the scratch buffer for the split-K slabs (a side stream must not share slot 0)."""
    n, h, wd, cin = x.shape
    cout = dy.shape[3] if cout is None else cout
    dw = torch.empty((cout, 64) if cin == 4 else (cout, ksize, ksize, cin), dtype=torch.float32, device=x.device)
    nws = lib().conv2d_bwd_weight_ws(n, h, wd, cin, cout, ksize, stride)
    ws = scratch(nws, x.device, slot=slot) if nws > 0 else None
    geom = conv_geom(x.device, n, h, wd, ksize, stride)
    lib().conv2d_bwd_weight(x.data_ptr(), x.stride(2), dy.data_ptr(), dy.stride(2), dw.data_ptr(), _p(ws), geom.data_ptr(),
                            n, h, wd, cin, cout, ksize, stride, _s())
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


def wgrad_on_side(x, dy, ksize, stride, wshape):
    """Launch the weight gradient (+ its OHWI->OIHW conversion) on the side stream.  Returns the OIHW
    gradient; the caller must make the main stream wait for side_stream() before the result is consumed."""
    if not WGRAD_SIDE:
        dw = conv2d_bwd_weight(x, dy, ksize, stride)
        if wshape[0] != dw.shape[0]:
            dw = dw[:wshape[0]].contiguous()
        return weight_grad_to_oihw(dw, wshape)
    main = torch.cuda.current_stream()
    side = side_stream(x.device)
    side.wait_stream(main)                       # dy (and x) are produced on the main stream
    with torch.cuda.stream(side):
        dw = conv2d_bwd_weight(x, dy, ksize, stride, slot=3)
        if wshape[0] != dw.shape[0]:
            dw = dw[:wshape[0]].contiguous()
        out = weight_grad_to_oihw(dw, wshape)
    for t in (x, dy):
        t.record_stream(side)                    # keep the allocator from recycling them under the side kernels
    out.record_stream(main)
    return out


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


def scale_act(y, scale, shift, act, slope, residual=None, out=None):
    c = y.shape[-1]
    rows = y.numel() // c
    if out is None:
        out = torch.empty_like(y)
    _chk(y, "scale_act y"); _rows(out, "scale_act out")
    lib().scale_act(y.data_ptr(), _p(scale), _p(shift), act, float(slope), _p(residual), out.data_ptr(), rows, c,
                    out.stride(-2), _s())
    return out


def bn_act_bwd(y, dout, mean, invstd, gamma, beta, act, slope):
    """Returns (dy, dgamma, dbeta) for out = act(gamma*(y-mean)*invstd+beta) with batch statistics."""
    c = y.shape[-1]
    rows = y.numel() // c
    dev = y.device
    r = lib().channel_stats_rows(rows)
    part = scratch(r * 2 * c, dev, slot=0)
    lddo = dout.stride(-2)
    lib().bn_act_bwd_reduce(y.data_ptr(), dout.data_ptr(), lddo, mean.data_ptr(), invstd.data_ptr(), _p(gamma), _p(beta),
                            act, float(slope), rows, c, part.data_ptr(), _s())
    sums = torch.empty((2, c), dtype=torch.float32, device=dev)
    ws = scratch(lib().bn_ws(c), dev, slot=2)
    lib().bn_bwd_sums(part.data_ptr(), r, c, sums.data_ptr(), ws.data_ptr(), _s())
    dy = torch.empty_like(y)
    lib().bn_act_bwd_apply(y.data_ptr(), dout.data_ptr(), lddo, mean.data_ptr(), invstd.data_ptr(), _p(gamma), _p(beta),
                           act, float(slope), sums.data_ptr(), rows, rows, c, dy.data_ptr(), _s())
    return dy, sums[1], sums[0]


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
    E = torch.empty(lib().coattn_e_size(b, hw), dtype=torch.float32, device=dev)
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


# ---- scoring ------------------------------------------------------------------------------------------
def l2norm_score_fwd(x, q=None, rows_per_image=0, out=None):
    """x (...,c) rows (pixel stride = x.stride(-2)).  Returns (out, norm, score|None)."""
    c = x.shape[-1]
    rows = x.numel() // c
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _rows(x, "l2norm x"); _rows(out, "l2norm out")
    norm = torch.empty(rows, dtype=torch.float32, device=x.device)
    score = torch.empty(rows, dtype=torch.float32, device=x.device) if q is not None else None
    lib().l2norm_score_fwd(x.data_ptr(), x.stride(-2), out.data_ptr(), out.stride(-2), norm.data_ptr(), _p(q), _p(score),
                           rows, rows_per_image, c, _s())
    return out, norm, score


def l2norm_score_bwd(out, norm, dout, q, dscore, rows_per_image, want_dq=True):
    c = out.shape[-1]
    rows = out.numel() // c
    _rows(out, "l2norm_bwd out")
    if dout is not None:
        _rows(dout, "l2norm_bwd dout")
    dx = torch.empty(out.shape, dtype=torch.float32, device=out.device)
    dq = None
    if q is not None and dscore is not None and want_dq:
        dq = torch.empty((rows // rows_per_image, c), dtype=torch.float32, device=out.device)
    lib().l2norm_score_bwd(out.data_ptr(), out.stride(-2), norm.data_ptr(), _p(dout), 0 if dout is None else dout.stride(-2),
                           _p(q), _p(dscore), dx.data_ptr(), c, _p(dq), rows, rows_per_image, c, _s())
    return dx, dq


# ---- data movers ---------------------------------------------------------------------------------------
def upsample2_into(src, dst_view):
    """src (N,h,w,c) -> dst_view (N,2h,2w,c) (a channel slice of a wider NHWC buffer)."""
    n, h, w, c = src.shape
    lib().upsample2_nhwc(src.data_ptr(), src.stride(2), dst_view.data_ptr(), dst_view.stride(2), n, h, w, c, _s())


def upsample2_bwd(ddst_view, dsrc, accumulate):
    n, h, w, c = dsrc.shape
    lib().upsample2_nhwc_bwd(ddst_view.data_ptr(), ddst_view.stride(2), dsrc.data_ptr(), dsrc.stride(2), n, h, w, c,
                             int(accumulate), _s())


def copy_slice(src_view, dst_view, accumulate=False):
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
