"""GPU parity of every C-ABI kernel against a plain PyTorch fp32/fp64 CPU reference of the same op.
All calls go through dcnet_amd.ops -> ctypes -> libdcnet_hip.so."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _close(a, b, tol, name=""):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    ref = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * ref, f"{name}: max err {err:.3e} vs tol {tol * ref:.3e}"


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # n, h, w, cin, cout, k, stride
    (2, 13, 13, 64, 128, 3, 1),
    (2, 13, 13, 128, 64, 1, 1),
    (1, 26, 26, 32, 64, 3, 2),
    (2, 16, 20, 64, 32, 1, 1),       # Cout 32 -> 256x32 tile
    (1, 9, 11, 96, 160, 3, 1),       # ragged M and Cout (masks)
    (3, 8, 8, 256, 255, 1, 1),       # Cout not a multiple of anything
    (1, 32, 32, 3, 32, 3, 1),        # stem (c4 path)
    (2, 7, 7, 512, 15, 1, 1),        # bbox head width
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_plain(dev, case):
    from dcnet_amd import ops
    n, h, w, cin, cout, k, s = case
    x = _rand(n, cin, h, w, seed=1); wt = _rand(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5)
    ref = F.conv2d(x.double(), wt.double(), None, s, (k - 1) // 2).float()
    xd = ops.nchw_to_nhwc(x.to(dev), 4 if cin == 3 else None)
    wd = ops.weight_to_ohwi(wt.to(dev))
    y, _ = ops.conv2d_fwd(xd, wd, k, s)
    _close(ops.nhwc_to_nchw(y), ref, 2e-5, "conv fwd")


def test_conv2d_fwd_fused_epilogue_and_stats(dev):
    from dcnet_amd import ops
    n, h, w, cin, cout, k, s = 2, 14, 14, 64, 96, 3, 1
    x = _rand(n, cin, h, w, seed=3); wt = _rand(cout, cin, k, k, seed=4, scale=(cin * 9) ** -0.5)
    scale = _rand(cout, seed=5).abs() + 0.5; shift = _rand(cout, seed=6); res = _rand(n, cout, h, w, seed=7)
    raw = F.conv2d(x.double(), wt.double(), None, s, 1)
    ref = F.leaky_relu(raw * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1), 0.1) + res.double()
    xd = ops.nchw_to_nhwc(x.to(dev)); wd = ops.weight_to_ohwi(wt.to(dev))
    y, stats = ops.conv2d_fwd(xd, wd, k, s, scale.to(dev), shift.to(dev), ops.ACT_LEAKY, 0.1,
                              residual=ops.nchw_to_nhwc(res.to(dev)), want_stats=True)
    _close(ops.nhwc_to_nchw(y), ref.float(), 2e-5, "fused epilogue")
    tot = stats.sum(0).cpu().double()
    _close(tot[0], raw.sum((0, 2, 3)), 1e-5, "stats sum")
    _close(tot[1], (raw * raw).sum((0, 2, 3)), 1e-5, "stats sumsq")


@pytest.mark.parametrize("n,h,w,cout", [(1, 32, 32, 32), (3, 19, 23, 32), (2, 40, 56, 16), (1, 5, 3, 32), (2, 64, 64, 24)])
def test_stem_direct_kernel(dev, n, h, w, cout):
    """csrc/stem.hip (the 4-channel 3x3 stem directly on the vector ALU, weights through the scalar cache) against fp64 and
    against the implicit-GEMM c4 tile it replaces: plain, with BatchNorm partial sums (same [row][2][Co] layout: the caller
    sizes the buffer before it knows which kernel runs), with the folded scale / shift / LeakyReLU epilogue and the abs-max
    word, into a wider destination.  Image borders and the ragged last block of 256 pixels matter here."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    x = _rand(n, 3, h, w, seed=61); wt = _rand(cout, 3, 3, 3, seed=62, scale=27 ** -0.5)
    scale = (_rand(cout, seed=63).abs() + 0.5); shift = _rand(cout, seed=64)
    raw = F.conv2d(x.double(), wt.double(), None, 1, 1)
    ref = F.leaky_relu(raw * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1), 0.1)
    xd = ops.nchw_to_nhwc(x.to(dev), 4); wd = ops.weight_to_ohwi(wt.to(dev))

    def run():
        out = {}
        out["y"], out["stats"] = ops.conv2d_fwd(xd, wd, 3, 1, want_stats=True)
        am = ops.amax_slot(dev)
        buf = torch.zeros(n, h, w, cout + 8, device=dev)
        ops.conv2d_fwd(xd, wd, 3, 1, scale.to(dev), shift.to(dev), ops.ACT_LEAKY, 0.1, out=buf[..., 4:4 + cout], amax_out=am)
        out["buf"] = buf; out["amax"] = am.clone()
        return out

    try:
        lib().set_tuning(b"jstem", 0)
        old = run()
        lib().set_tuning(b"jstem", 1)
        new = run()
    finally:
        lib().set_tuning(b"jstem", 1)
    for r, name in ((old, "c4 tile"), (new, "direct")):
        _close(ops.nhwc_to_nchw(r["y"]), raw, 2e-5, f"stem {name}")
        tot = r["stats"].sum(0).cpu().double()
        _close(tot[0], raw.sum((0, 2, 3)), 1e-5, f"stem stats sum {name}")
        _close(tot[1], (raw * raw).sum((0, 2, 3)), 1e-5, f"stem stats sumsq {name}")
        _close(_nchw(r["buf"][..., 4:4 + cout]), ref, 2e-5, f"stem epilogue {name}")
        assert float(r["buf"][..., :4].abs().max()) == 0 and float(r["buf"][..., 4 + cout:].abs().max()) == 0
        assert float(r["amax"].view(torch.float32).max()) == float(r["buf"].abs().max())
    assert new["stats"].shape == old["stats"].shape
    _close(new["stats"].view(-1, 2 * cout), old["stats"].view(-1, 2 * cout), 1e-5, "stem stats rows")     # row by row
    if cout % 4 == 0:
        assert not torch.equal(new["y"], old["y"]), "the direct stem kernel did not run"
    # a destination that is not 16-byte aligned stays on the implicit-GEMM tile (and is still right)
    odd = torch.zeros(n, h, w, cout + 4, device=dev)
    ops.conv2d_fwd(xd, wd, 3, 1, out=odd[..., 1:1 + cout])
    _close(_nchw(odd[..., 1:1 + cout]), raw, 2e-5, "stem, unaligned destination")
    assert float(odd[..., 0].abs().max()) == 0 and float(odd[..., 1 + cout:].abs().max()) == 0


def test_conv2d_fwd_into_concat_slice(dev):
    from dcnet_amd import ops
    n, h, w, cin, cout = 2, 10, 10, 64, 64
    x = _rand(n, cin, h, w, seed=8); wt = _rand(cout, cin, 1, 1, seed=9, scale=cin ** -0.5)
    buf = torch.zeros(n, h, w, 192, device=dev)
    ops.conv2d_fwd(ops.nchw_to_nhwc(x.to(dev)), ops.weight_to_ohwi(wt.to(dev)), 1, 1, out=buf[..., 64:128])
    ref = F.conv2d(x, wt)
    _close(_nchw(buf[..., 64:128]), ref, 2e-5, "slice")
    assert float(buf[..., :64].abs().max()) == 0 and float(buf[..., 128:].abs().max()) == 0


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[3] != 3 and c[4] % 32 == 0])
def test_conv2d_bwd_data(dev, case):
    from dcnet_amd import ops
    n, h, w, cin, cout, k, s = case
    x = _rand(n, cin, h, w, seed=1).double().requires_grad_(True)
    wt = _rand(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5)
    y = F.conv2d(x, wt.double(), None, s, (k - 1) // 2)
    dy = _rand(*y.shape, seed=11)
    y.backward(dy.double())
    dx = ops.conv2d_bwd_data(ops.nchw_to_nhwc(dy.to(dev)), ops.weight_to_ohwi(wt.to(dev)), (h, w), k, s)
    _close(ops.nhwc_to_nchw(dx), x.grad.float(), 2e-5, "dgrad")
    # accumulate form
    base = _rand(n, cin, h, w, seed=12)
    dx2 = ops.nchw_to_nhwc(base.to(dev))
    ops.conv2d_bwd_data(ops.nchw_to_nhwc(dy.to(dev)), ops.weight_to_ohwi(wt.to(dev)), (h, w), k, s, out=dx2, accumulate=True)
    _close(ops.nhwc_to_nchw(dx2), x.grad.float() + base, 2e-5, "dgrad accumulate")


WGRAD_CASES = CONV_CASES[:5] + [CONV_CASES[6], (4, 52, 52, 32, 64, 3, 1), (2, 20, 20, 128, 128, 3, 1), (2, 12, 12, 32, 32, 1, 1),
                                (2, 12, 12, 64, 32, 3, 1), (2, 12, 12, 32, 128, 1, 1)]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv2d_bwd_weight(dev, case):
    from dcnet_amd import ops
    n, h, w, cin, cout, k, s = case
    x = _rand(n, cin, h, w, seed=1)
    wt = (_rand(cout, cin, k, k, seed=2) * 0.1).double().requires_grad_(True)
    y = F.conv2d(x.double(), wt, None, s, (k - 1) // 2)
    dy = _rand(*y.shape, seed=13)
    y.backward(dy.double())
    xd = ops.nchw_to_nhwc(x.to(dev), 4 if cin == 3 else None)
    dw = ops.conv2d_bwd_weight(xd, ops.nchw_to_nhwc(dy.to(dev)), k, s)
    got = ops.weight_grad_to_oihw(dw, (cout, cin, k, k))
    _close(got, wt.grad.float(), 3e-5, "wgrad")


@pytest.mark.parametrize("n,h,w,c", [(3, 9, 7, 96), (2, 5, 5, 4), (1, 33, 31, 68), (4, 64, 64, 32), (2, 416, 416, 8), (4, 416, 416, 8), (5, 13, 13, 1024)])
def test_batchnorm_train_fwd_bwd(dev, n, h, w, c):
    """(2,416,416,8): 2704 partial rows -> the one-launch reduce; (4,416,416,8): 5408 partial rows -> the two-stage
    reduce; the others: ragged channel blocks and tiny / wide channel counts."""
    from dcnet_amd import ops
    y = (_rand(n, c, h, w, seed=20) * 2 + 0.5)
    gamma = _rand(c, seed=21).abs() + 0.5; beta = _rand(c, seed=22)
    rm = _rand(c, seed=23); rv = _rand(c, seed=24).abs() + 0.5
    res = _rand(n, c, h, w, seed=25)
    yd = y.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.clone().double(), rv.clone().double()
    out_ref = F.leaky_relu(F.batch_norm(yd, rm_ref, rv_ref, gd, bd, True, 0.1, 1e-5), 0.1) + res.double()
    dout = _rand(n, c, h, w, seed=26)
    out_ref.backward(dout.double())

    y_nhwc = ops.nchw_to_nhwc(y.to(dev))
    stats = ops.channel_stats(y_nhwc.view(-1, c))
    rmd, rvd = rm.to(dev), rv.to(dev)
    mi = ops.bn_finalize(stats, n * h * w, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rmd, rvd)
    out = ops.scale_act(y_nhwc, mi[2], mi[3], ops.ACT_LEAKY, 0.1, residual=ops.nchw_to_nhwc(res.to(dev)))
    _close(ops.nhwc_to_nchw(out), out_ref.float(), 2e-5, "bn fwd")
    _close(rmd, rm_ref.float(), 1e-5, "running_mean"); _close(rvd, rv_ref.float(), 1e-5, "running_var")
    dy, dgamma, dbeta = ops.bn_act_bwd(y_nhwc, ops.nchw_to_nhwc(dout.to(dev)), mi[0], mi[1], gamma.to(dev), beta.to(dev),
                                       ops.ACT_LEAKY, 0.1)
    _close(ops.nhwc_to_nchw(dy), yd.grad.float(), 3e-5, "bn dy")
    _close(dgamma, gd.grad.float(), 3e-5, "dgamma"); _close(dbeta, bd.grad.float(), 3e-5, "dbeta")


def test_bn_fold_and_act_bwd(dev):
    from dcnet_amd import ops
    c = 64
    gamma = _rand(c, seed=1).abs() + 0.5; beta = _rand(c, seed=2); rm = _rand(c, seed=3); rv = _rand(c, seed=4).abs() + 0.1
    ss = ops.bn_fold(gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev))
    sc = gamma / torch.sqrt(rv + 1e-5)
    _close(ss[0], sc, 1e-6); _close(ss[1], beta - rm * sc, 1e-6)
    o = _rand(5, 6, c, seed=5); d = _rand(5, 6, c, seed=6)
    got = ops.act_bwd(o.to(dev), d.to(dev), 0.1)
    _close(got, torch.where(o > 0, d, d * 0.1), 1e-6)


@pytest.mark.parametrize("b,g,c", [(2, 8, 64), (1, 13, 512), (3, 5, 96), (8, 13, 128), (2, 26, 128), (1, 52, 128),   # (the last three: >= 1024 rows, the f16-split GEMMs; 52x52: the TN tile too)
                                   (2, 26, 256), (1, 23, 256), (2, 52, 512)])      # all nine products on pre-split operands (csrc/gemm3.hip); 23 x 23: odd row counts
def test_coattn_fwd_bwd(dev, b, g, c):
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    hw = g * g
    lib().prof_enable(1)
    f1 = F.normalize(_rand(b, hw, c, seed=30), dim=2); f2 = F.normalize(_rand(b, hw, c, seed=31), dim=2)
    a = f1.double().requires_grad_(True); bb = f2.double().requires_grad_(True)
    A = torch.bmm(a, bb.transpose(1, 2))
    o1 = torch.bmm(F.softmax(A * 10, dim=2), bb)                       # f1_attn[i] = sum_j softmax_j
    o2 = torch.bmm(F.softmax(A * 10, dim=1).transpose(1, 2), a)        # f2_attn[j] = sum_i softmax_i
    d1 = _rand(b, hw, c, seed=32); d2 = _rand(b, hw, c, seed=33)
    (o1 * d1.double()).sum().backward(retain_graph=True)
    (o2 * d2.double()).sum().backward()
    f1d, f2d = f1.to(dev), f2.to(dev)
    cat = torch.zeros(2, b, hw, 2 * c, device=dev)                     # outputs land in a slice (ldo = 2c)
    out1, out2 = cat[0, :, :, c:], cat[1, :, :, c:]
    E, rc = ops.coattn_fwd(f1d, f2d, out1, out2, 10.0)
    _close(out1, o1.float(), 2e-5, "f1_attn"); _close(out2, o2.float(), 2e-5, "f2_attn")
    base1 = _rand(b, hw, c, seed=34); base2 = _rand(b, hw, c, seed=35)
    g1, g2 = base1.to(dev).clone(), base2.to(dev).clone()
    ops.coattn_bwd(f1d, f2d, d1.to(dev), d2.to(dev), out1, out2, E, rc, g1, g2, True, 10.0)
    _close(g1, a.grad.float() + base1, 5e-5, "d_f1"); _close(g2, bb.grad.float() + base2, 5e-5, "d_f2")
    # inference form: only f1_attn
    o_only = torch.empty(b, hw, c, device=dev)
    ops.coattn_fwd(f1d, f2d, o_only, None, 10.0)
    _close(o_only, o1.float(), 2e-5, "f1_attn only")
    lib().prof_enable(0)
    assert _prof_launches(40) == (11 if (hw >= 512 and c >= 256) else 0), "which engine ran the products"


def test_l2norm_score_fwd_bwd(dev):
    from dcnet_amd import ops
    n, hw, c = 3, 37, 512
    x = _rand(n, hw, c, seed=40); q = F.normalize(_rand(n, c, seed=41), dim=1)
    xd = x.double().requires_grad_(True); qd = q.double().requires_grad_(True)
    o = F.normalize(xd, dim=2); sc = (o * qd.unsqueeze(1)).sum(2)
    do = _rand(n, hw, c, seed=42); ds = _rand(n, hw, seed=43)
    ((o * do.double()).sum() + (sc * ds.double()).sum()).backward()
    out, norm, score, _ = ops.l2norm_score_fwd(x.to(dev), q.to(dev), hw)
    _close(out, o.float(), 1e-6, "normalize"); _close(score.view(n, hw), sc.float(), 1e-5, "score")
    dx, dq = ops.l2norm_score_bwd(out, norm, do.to(dev), q.to(dev), ds.to(dev).view(-1), hw)
    _close(dx, xd.grad.float(), 2e-5, "dx"); _close(dq, qd.grad.float(), 2e-5, "dq")
    out2, norm2, none, none2 = ops.l2norm_score_fwd(x.to(dev))
    assert none is None and none2 is None
    _close(out2, o.float(), 1e-6)
    dx2, _ = ops.l2norm_score_bwd(out2, norm2, do.to(dev), None, None, 0)
    xd2 = x.double().requires_grad_(True)
    (F.normalize(xd2, dim=2) * do.double()).sum().backward()
    _close(dx2, xd2.grad.float(), 2e-5, "dx (no score)")


def test_layout_and_movers(dev):
    from dcnet_amd import ops
    x = _rand(2, 5, 7, 9, seed=50)
    xd = ops.nchw_to_nhwc(x.to(dev), 8)
    assert xd.shape == (2, 7, 9, 8)
    _close(xd[..., :5], x.permute(0, 2, 3, 1), 0.0); assert float(xd[..., 5:].abs().max()) == 0
    _close(ops.nhwc_to_nchw(xd, 5), x, 0.0)
    w = _rand(6, 40, 3, 3, seed=51)
    wo = ops.weight_to_ohwi(w.to(dev))
    assert wo.shape == (6, 3, 3, 64)
    _close(wo[..., :40], w.permute(0, 2, 3, 1), 0.0)
    _close(ops.weight_grad_to_oihw(wo, (6, 40, 3, 3)), w, 0.0)
    s = _rand(2, 4, 5, 8, seed=52).to(dev)
    buf = torch.zeros(2, 8, 10, 24, device=dev)
    ops.upsample2_into(s, buf[..., 8:16])
    ref = s.repeat_interleave(2, 1).repeat_interleave(2, 2)
    _close(buf[..., 8:16], ref, 0.0)
    g = _rand(2, 8, 10, 24, seed=53).to(dev)
    ds = torch.ones(2, 4, 5, 8, device=dev)
    ops.upsample2_bwd(g[..., 8:16], ds, True)
    gr = g[..., 8:16].reshape(2, 4, 2, 5, 2, 8).sum((2, 4)) + 1
    _close(ds, gr, 1e-6)
    d = torch.ones(2, 8, 10, 8, device=dev)
    ops.copy_slice(g[..., 16:24], d, True)
    _close(d, g[..., 16:24] + 1, 1e-6)


def test_linear_bn1d_functions(dev):
    from dcnet_amd.functions import BatchNormRowsAct, LinearAct
    torch.manual_seed(0)
    lin = torch.nn.Linear(1024, 512); bn = torch.nn.BatchNorm1d(512)
    x = _rand(64, 1024, seed=60)
    xr = x.double().requires_grad_(True)
    lr = torch.nn.Linear(1024, 512).double(); lr.load_state_dict(lin.state_dict()); br = torch.nn.BatchNorm1d(512).double()
    ref = torch.relu(br(lr(xr)))
    g = _rand(64, 512, seed=61)
    ref.backward(g.double())
    lin, bn = lin.to(dev), bn.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    z = LinearAct.apply(xd, lin.weight, lin.bias, False)
    out = BatchNormRowsAct.apply(z, bn.weight, bn.bias, bn, True, True)
    out.backward(g.to(dev))
    _close(out, ref.float(), 2e-5, "linear+bn1d fwd")
    _close(xd.grad, xr.grad.float(), 5e-5, "dx"); _close(lin.weight.grad, lr.weight.grad.float(), 5e-5, "dW")
    _close(lin.bias.grad, lr.bias.grad.float(), 5e-5, "db")
    _close(bn.weight.grad, br.weight.grad.float(), 5e-5, "dgamma"); _close(bn.running_var, br.running_var.float(), 1e-5, "rv")
    # Linear + ReLU
    x2 = _rand(40, 512, seed=62); l2 = torch.nn.Linear(512, 512)
    x2r = x2.double().requires_grad_(True); l2r = torch.nn.Linear(512, 512).double(); l2r.load_state_dict(l2.state_dict())
    r2 = torch.relu(l2r(x2r)); g2 = _rand(40, 512, seed=63); r2.backward(g2.double())
    l2 = l2.to(dev); x2d = x2.to(dev).requires_grad_(True)
    o2 = LinearAct.apply(x2d, l2.weight, l2.bias, True); o2.backward(g2.to(dev))
    _close(o2, r2.float(), 2e-5); _close(x2d.grad, x2r.grad.float(), 5e-5); _close(l2.weight.grad, l2r.weight.grad.float(), 5e-5)


@pytest.mark.parametrize("ragged,n", [(False, 6), (True, 6), (True, 64), (True, 130)])
def test_bilstm_matches_torch_packed_lstm(dev, ragged, n):
    """The persistent BiLSTM (csrc/lstm.hip: one launch per pass, 128 co-resident workgroups exchanging h_t through
    global memory) against nn.LSTM on a packed batch in fp64; n = 64 fills every lane, n = 130 needs three row chunks."""
    from dcnet_amd import ops
    from dcnet_amd.functions import BiLSTM
    torch.manual_seed(1)
    L, I, H = 20, 512, 512
    ref = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True).double()
    x = _rand(n, L, I, seed=70)
    lengths = (torch.tensor([20, 7, 20, 13, 1, 20] * 22)[:n] if ragged else torch.full((n,), L))
    xr = x.double().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lengths, batch_first=True, enforce_sorted=False)
    yr, _ = ref(packed)
    yr, _ = torch.nn.utils.rnn.pad_packed_sequence(yr, batch_first=True, total_length=L)
    g = _rand(n, L, 2 * H, seed=71)
    (yr * g.double()).sum().backward()
    params = [p.detach().float().to(dev).requires_grad_(True) for p in
              (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0,
               ref.weight_ih_l0_reverse, ref.weight_hh_l0_reverse, ref.bias_ih_l0_reverse, ref.bias_hh_l0_reverse)]
    xd = x.to(dev).requires_grad_(True)
    y = BiLSTM.apply(xd, lengths.to(dev), *params)
    (y * g.to(dev)).sum().backward()
    assert not ops.bilstm_sync_error(dev), "a bounded spin of the persistent kernel gave up"
    _close(y, yr.float(), 2e-5, "lstm out")
    _close(xd.grad, xr.grad.float(), 1e-4, "lstm dx")
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
             "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    for p, nm in zip(params, names):
        _close(p.grad, getattr(ref, nm).grad.float(), 1e-4, nm)


def test_bilstm_error_word_is_seen_from_another_stream(dev):
    """The model launches the persistent BiLSTM on its language side stream and callers check on the main stream: the check
    must look at the sync buffer of EVERY stream of the device (round-2 advisor finding: it created a fresh zeroed buffer for
    the caller's stream and always reported 'no error')."""
    from dcnet_amd import ops
    from dcnet_amd.functions import BiLSTM
    torch.manual_seed(2)
    L, I, H, n = 20, 512, 512, 4
    params = [(torch.randn(s_) * 0.05).to(dev) for s_ in ((4 * H, I), (4 * H, H), (4 * H,), (4 * H,)) * 2]
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        BiLSTM.apply(_rand(n, L, I, seed=72).to(dev), torch.full((n,), L, device=dev), *params)
    key = (dev.index if dev.index is not None else 0, side.cuda_stream)
    assert key in ops._lstm_sync, "the launch on the side stream must own a sync buffer"
    assert not ops.bilstm_sync_error(dev)
    ops._lstm_sync[key][8] = 1                        # what a timed-out spin leaves behind
    assert ops.bilstm_sync_error(dev), "a sticky error word on another stream's buffer must be reported"
    with pytest.raises(ops.DcnError):
        ops.check_bilstm(dev)
    ops._lstm_sync[key][8] = 0


def test_location_module_core(dev):
    from dcnet_amd.functions import LocModule
    n, p, c = 3, 341, 512
    E = F.normalize(torch.relu(_rand(p, 8, seed=80)) + 0.01, dim=1)
    M = _rand(n, 8, c, seed=81) * 0.3; b = _rand(c, seed=82) * 0.1; q = F.normalize(_rand(n, c, seed=83), dim=1)
    Er, Mr, br, qr = (t.double().requires_grad_(True) for t in (E, M, b, q))
    rel = torch.relu(torch.matmul(Er.unsqueeze(0), Mr) + br)              # (n,p,c)
    ref = (F.normalize(rel, dim=2) * qr.unsqueeze(1)).sum(2)
    g = _rand(n, p, seed=84)
    (ref * g.double()).sum().backward()
    Ed, Md, bd, qd = (t.to(dev).requires_grad_(True) for t in (E, M, b, q))
    out = LocModule.apply(Ed, Md, bd, qd)
    (out * g.to(dev)).sum().backward()
    _close(out, ref.float(), 1e-5, "loc fwd")
    _close(Ed.grad, Er.grad.float(), 5e-5, "dE"); _close(Md.grad, Mr.grad.float(), 5e-5, "dM")
    _close(bd.grad, br.grad.float(), 5e-5, "db"); _close(qd.grad, qr.grad.float(), 5e-5, "dq")


def test_fusion_conv_fold_matches_concat_form(dev):
    """FusionConvBNAct == ConvBatchNormReLU over the concat [corr | tile(flang) | coord] (DCNet_model.py:491-505)."""
    from dcnet_amd.functions import FusionConvBNAct
    from dcnet_amd.model import generate_coord_nhwc
    torch.manual_seed(3)
    n, g, e, co = 3, 7, 512, 512
    corr = F.normalize(_rand(n, e, g, g, seed=90), dim=1); flang = F.normalize(_rand(n, e, seed=91), dim=1)
    coord = generate_coord_nhwc(g, g, "cpu")                                   # (g,g,8)
    conv = torch.nn.Conv2d(2 * e + 8, co, 1, bias=False); bn = torch.nn.BatchNorm2d(co, momentum=0.999)
    torch.nn.init.normal_(conv.weight, std=0.03)
    cr, fr = corr.double().requires_grad_(True), flang.double().requires_grad_(True)
    convr = torch.nn.Conv2d(2 * e + 8, co, 1, bias=False).double(); convr.load_state_dict(conv.state_dict())
    bnr = torch.nn.BatchNorm2d(co, momentum=0.999).double()
    cat = torch.cat([cr, fr.view(n, e, 1, 1).repeat(1, 1, g, g), coord.permute(2, 0, 1).unsqueeze(0).repeat(n, 1, 1, 1).double()], 1)
    ref = torch.relu(bnr(convr(cat)))
    gup = _rand(n, co, g, g, seed=92)
    ref.backward(gup.double())
    conv, bn = conv.to(dev), bn.to(dev)
    cd = corr.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True); fd = flang.to(dev).requires_grad_(True)
    out, _ = FusionConvBNAct.apply(cd, fd, coord.to(dev), conv.weight, bn.weight, bn.bias, bn, True)
    out.backward(gup.permute(0, 2, 3, 1).contiguous().to(dev))
    _close(out.permute(0, 3, 1, 2), ref.float(), 3e-5, "fusion fwd")
    _close(cd.grad.permute(0, 3, 1, 2), cr.grad.float(), 5e-5, "dcorr"); _close(fd.grad, fr.grad.float(), 5e-5, "dflang")
    _close(conv.weight.grad, convr.weight.grad.float(), 5e-5, "dW"); _close(bn.weight.grad, bnr.weight.grad.float(), 5e-5, "dgamma")
    _close(bn.running_var, bnr.running_var.float(), 1e-5, "running_var")


# ---- the split-bf16 matrix pipe ("precision" 1, the default on 128x128 tiles) ---------------------------
SPLIT_CASES = [
    # n, h, w, cin, cout, k, stride      (>= 1024 output rows and Cout > 64 select the split tiles by default)
    (2, 26, 26, 64, 128, 3, 1),
    (1, 40, 36, 96, 160, 3, 1),      # ragged M tile, ragged Cout tile
    (2, 32, 32, 128, 252, 1, 1),     # Cout tail of 124
    (2, 52, 52, 128, 256, 3, 2),     # stride 2 (dgrad: four parity classes)
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_split_pipe_is_fp32_accurate(dev, case):
    """Forward, data gradient and weight gradient on the split matrix pipes — mode 4 (the default): two f16 pieces per
    operand with per-tensor power-of-two scales, three MFMAs per product; mode 1: three bf16 pieces, six MFMAs — against
    an fp64 reference, next to the native fp32 MFMA kernels (mode 0) on the same inputs: a split path must be at least as
    close (factor 2 slack + 1e-6) — it is a different instruction sequence, not a different precision.  The inputs carry
    a wide dynamic range (a few elements 1e4 x larger, many 1e-4 x smaller) so that the scaling is exercised."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout, k, st = case
    x = _rand(n, h, w, cin, seed=1)
    spread = torch.exp(2.5 * _rand(n, h, w, cin, seed=7))          # log-normal magnitudes: ~1e-4 .. 1e4
    x = (x * spread).to(dev)
    wt = (_rand(cout, k, k, cin, seed=2) / (cin * k * k) ** 0.5).to(dev)
    xd = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wd = wt.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    yd = F.conv2d(xd, wd, stride=st, padding=(k - 1) // 2)
    dy = (_rand(*yd.shape, seed=3) * 1e-6 * torch.exp(2.0 * _rand(*yd.shape, seed=8))).permute(0, 2, 3, 1).contiguous().to(dev)   # tiny gradients
    yd.backward(dy.permute(0, 3, 1, 2).double().cpu())
    ref = {"fwd": yd.detach().permute(0, 2, 3, 1), "dgrad": xd.grad.permute(0, 2, 3, 1), "wgrad": wd.grad.permute(0, 2, 3, 1)}
    cout_p = (cout + 31) // 32 * 32                     # dgrad contracts over Cout: padded filter bank / dy
    wt_p = torch.zeros(cout_p, k, k, cin, device=dev); wt_p[:cout] = wt
    dy_p = torch.zeros(*dy.shape[:3], cout_p, device=dev); dy_p[..., :cout] = dy
    err = {}
    try:
        for mode in (0, 1, 4):
            lib().set_tuning(b"precision", mode)
            got = {"fwd": ops.conv2d_fwd(x, wt, k, st)[0],
                   "dgrad": ops.conv2d_bwd_data(dy_p, wt_p, (h, w), k, st),
                   "wgrad": ops.conv2d_bwd_weight(x, dy, k, st)}
            for name, t in got.items():
                err[(name, mode)] = float((t.double().cpu().reshape(ref[name].shape) - ref[name]).abs().max())
    finally:
        lib().set_tuning(b"precision", 4)
    for name in ref:
        scale = float(ref[name].abs().max())
        for mode in (1, 4):
            assert err[(name, mode)] <= 2 * err[(name, 0)] + 1e-6 * scale, (name, mode, err)
            assert err[(name, mode)] <= 3e-5 * scale, (name, mode, err)


def test_f16_split_is_live_and_survives_extreme_operands(dev):
    """Mode 4 really runs the f16 kernels (its result differs in the last bits from the bf16 three-piece split), and the
    abs-max scaling keeps huge (1e30), tiny (1e-30) and all-zero operands finite and accurate."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout = 2, 32, 32, 128, 128
    base = _rand(n, h, w, cin, seed=11)
    wt = (_rand(cout, 3, 3, cin, seed=12) / 34).to(dev)
    try:
        for mag in (1.0, 1e30, 1e-30):
            x = (base * mag).to(dev)
            ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.permute(0, 3, 1, 2).double().cpu(), padding=1).permute(0, 2, 3, 1)
            lib().set_tuning(b"precision", 4)
            y4 = ops.conv2d_fwd(x, wt, 3, 1)[0]
            lib().set_tuning(b"precision", 1)
            y1 = ops.conv2d_fwd(x, wt, 3, 1)[0]
            assert torch.isfinite(y4).all()
            _close(y4 / mag, ref / mag, 3e-6, f"f16 split at magnitude {mag}")
            if mag == 1.0:
                assert not torch.equal(y4, y1), "precision 4 did not select the f16-split kernel"
        lib().set_tuning(b"precision", 4)
        z = ops.conv2d_fwd(torch.zeros_like(base).to(dev), wt, 3, 1)[0]
        assert float(z.abs().max()) == 0.0
    finally:
        lib().set_tuning(b"precision", 4)


def test_split_pipe_forced_on_every_nt_tile(dev):
    """dcn_set_tuning("split", 16|32) pushes the narrow tiles (128x64, 256x32) and both K-steps through the split
    loop as well — exercised here so the knob used by tools/bench_convs.py stays correct."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    x = _rand(2, 20, 20, 64, seed=4).to(dev)
    try:
        for cout in (32, 64, 128):
            wt = (_rand(cout, 3, 3, 64, seed=5) / 24).to(dev)
            ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.permute(0, 3, 1, 2).double().cpu(), padding=1).permute(0, 2, 3, 1)
            for mode in (16, 32):
                lib().set_tuning(b"precision", 1)          # the forced split is the bf16 three-piece one
                lib().set_tuning(b"split", mode)
                y = ops.conv2d_fwd(x, wt, 3, 1)[0]
                _close(y, ref, 1e-5, f"cout {cout} split {mode}")
    finally:
        lib().set_tuning(b"split", 0)
        lib().set_tuning(b"precision", 4)


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_bf16_operand_mode_matches_its_exact_model(dev, case):
    """precision "bf16" (BASELINE.json configs[2]; builder-defined): operands rounded to bf16
    (nearest even), products exact, fp32 accumulation.  Its exact model is the same convolution on the
    bf16-rounded operands, so the kernels must match THAT to accumulation-order accuracy (3e-5), and differ from
    the unrounded fp64 result by about 2^-9 relative, not more (tolerance 1e-2 of the output scale)."""
    from dcnet_amd import ops
    n, h, w, cin, cout, k, st = case
    x = _rand(n, h, w, cin, seed=1).to(dev)
    wt = (_rand(cout, k, k, cin, seed=2) / (cin * k * k) ** 0.5).to(dev)
    rb = lambda t: t.bfloat16().double()                        # round to nearest even, as v_cvt_pk_bf16_f32
    dy = None
    refs = {}
    for tag, f in (("model", rb), ("exact", lambda t: t.double())):
        xd = f(x.cpu()).permute(0, 3, 1, 2).requires_grad_(True)
        wd = f(wt.cpu()).permute(0, 3, 1, 2).requires_grad_(True)
        yd = F.conv2d(xd, wd, stride=st, padding=(k - 1) // 2)
        if dy is None:
            dy = (_rand(*yd.shape, seed=3) / 8).permute(0, 2, 3, 1).contiguous().to(dev)
        yd.backward(f(dy.cpu()).permute(0, 3, 1, 2))
        refs[tag] = {"fwd": yd.detach().permute(0, 2, 3, 1), "dgrad": xd.grad.permute(0, 2, 3, 1), "wgrad": wd.grad.permute(0, 2, 3, 1)}
    cout_p = (cout + 31) // 32 * 32
    wt_p = torch.zeros(cout_p, k, k, cin, device=dev); wt_p[:cout] = wt
    dy_p = torch.zeros(*dy.shape[:3], cout_p, device=dev); dy_p[..., :cout] = dy
    try:
        ops.set_precision("bf16")
        got = {"fwd": ops.conv2d_fwd(x, wt, k, st)[0],
               "dgrad": ops.conv2d_bwd_data(dy_p, wt_p, (h, w), k, st),
               "wgrad": ops.conv2d_bwd_weight(x, dy, k, st)}
    finally:
        ops.set_precision("fp32")
    modes = {}
    for name, t in got.items():
        t = t.double().cpu().reshape(refs["model"][name].shape)
        scale = max(1.0, float(refs["exact"][name].abs().max()))
        e_model = float((t - refs["model"][name]).abs().max())
        e_exact = float((t - refs["exact"][name]).abs().max())
        # tiles the dispatcher keeps on the fp32 pipe (narrow N, short grids) stay exact: every launch is one or the other
        assert min(e_model, e_exact) <= 3e-5 * scale, (name, e_model, e_exact, scale)
        modes[name] = "bf16" if e_model < e_exact else "fp32"
        if modes[name] == "bf16":
            assert 1e-5 * scale < e_exact <= 1e-2 * scale, (name, e_exact, scale)  # it IS reduced precision, by about 2^-9
    if cout >= 128 and n * (h // st) * (w // st) >= 1024:
        assert modes["fwd"] == "bf16", modes                     # the wide forward tile runs with bf16 operands


@pytest.mark.parametrize("m,n,k,kvalid", [(2048, 384, 256, 0), (1500, 132, 320, 300), (2704, 512, 2720, 2704)])
def test_gemm_nn_split_pipe(dev, m, n, k, kvalid):
    """out = a[M,K] @ b[K,N] with the B operand N-contiguous (the co-attention's E.f2 products): >= 1024 rows run on the
    split-bf16 pipe with transposed LDS fragments; must match fp64 like the fp32 MFMA kernel does (K tail via kvalid)."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    a = _rand(m, k, seed=1).to(dev); b = (_rand(k, n, seed=2) / k ** 0.5).to(dev)
    kv = kvalid or k
    ref = a.double().cpu()[:, :kv] @ b.double().cpu()[:kv]
    err = {}
    try:
        for mode in (0, 1):
            lib().set_tuning(b"nnsplit", mode)
            out = ops.gemm_nn(a, b, kvalid=kvalid)
            err[mode] = float((out.double().cpu() - ref).abs().max())
    finally:
        lib().set_tuning(b"nnsplit", 1)
    scale = max(1.0, float(ref.abs().max()))
    assert err[1] <= 2 * err[0] + 1e-6 * scale and err[1] <= 3e-5 * scale, err


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_fp8_operand_mode_matches_its_exact_model(dev, case):
    """precision "fp8" (BASELINE.json configs[4]; builder-defined): forward and data gradient with operands scaled by a
    per-tensor power of two (derived in the kernel from the tensors' abs-max words), rounded to OCP e4m3 (nearest even), multiplied exactly and accumulated in
    fp32.  Exact model = the same convolution on the scaled-rounded-unscaled operands (torch.float8_e4m3fn on the CPU):
    the wide tiles must match THAT to accumulation-order accuracy; tiles the dispatcher keeps on the fp32 pipe stay exact."""
    import math
    from dcnet_amd import ops
    n, h, w, cin, cout, k, st = case
    x = _rand(n, h, w, cin, seed=1).to(dev)
    wt = (_rand(cout, k, k, cin, seed=2) / (cin * k * k) ** 0.5).to(dev)

    def q8(t):                                                  # scale, round to e4m3, unscale — all exact but the rounding
        _, e = math.frexp(float(t.abs().max()))                 # max = f * 2^e, f in [0.5, 1)
        s = 2.0 ** (8 - e)                                      # the tile's own scale: the maximum lands in [128, 256) < 448
        return (t * s).to(torch.float8_e4m3fn).double() / s

    cout_p = (cout + 31) // 32 * 32
    wt_p = torch.zeros(cout_p, k, k, cin); wt_p[:cout] = wt.cpu()
    ho, wo = (h + 2 * ((k - 1) // 2) - k) // st + 1, (w + 2 * ((k - 1) // 2) - k) // st + 1
    dy_p = torch.zeros(n, ho, wo, cout_p); dy_p[..., :cout] = _rand(n, ho, wo, cout, seed=3) / 8
    refs = {}
    for tag, f in (("model", q8), ("exact", lambda t: t.double())):
        refs[tag] = {
            "fwd": F.conv2d(f(x.cpu()).permute(0, 3, 1, 2), f(wt.cpu()).permute(0, 3, 1, 2), stride=st, padding=(k - 1) // 2).permute(0, 2, 3, 1),
            "dgrad": torch.nn.grad.conv2d_input((n, cin, h, w), f(wt_p).permute(0, 3, 1, 2), f(dy_p).permute(0, 3, 1, 2),
                                                stride=st, padding=(k - 1) // 2).permute(0, 2, 3, 1)}
    try:
        ops.set_precision("fp8")
        got = {"fwd": ops.conv2d_fwd(x, wt, k, st)[0], "dgrad": ops.conv2d_bwd_data(dy_p.to(dev), wt_p.to(dev), (h, w), k, st)}
    finally:
        ops.set_precision("fp32")
    modes = {}
    for name, t in got.items():
        t = t.double().cpu()
        scale = max(1.0, float(refs["exact"][name].abs().max()))
        e_model = float((t - refs["model"][name]).abs().max()); e_exact = float((t - refs["exact"][name]).abs().max())
        # 1e-4: measured 4e-5 (the fp8 MFMA's own accumulation order/rounding; the bf16 mode meets 3e-5) against an
        # operand-rounding effect of 4e-2 — three orders of magnitude apart
        assert min(e_model, e_exact) <= 1e-4 * scale, (name, e_model, e_exact, scale)
        modes[name] = "fp8" if e_model < e_exact else "fp32"
        if modes[name] == "fp8":
            assert 1e-4 * scale < e_exact <= 0.2 * scale, (name, e_exact, scale)      # 3 mantissa bits: a few per cent
    if cout >= 128 and n * ho * wo >= 1024:
        assert modes["fwd"] == "fp8", modes


def test_fused_rmsprop_matches_torch(dev):
    """dcnet_amd.optim.RMSprop (one fused pass, dcn_rmsprop_step) against torch.optim.RMSprop on the same parameters and
    gradients: ragged sizes (vector body + scalar tail), two groups with different rates, weight decay, 3 steps;
    the state_dict loads into torch's optimiser and back."""
    from dcnet_amd.optim import RMSprop
    shapes = [(7,), (64, 33), (3, 3, 16, 5), (1,), (1024, 257), (40, 8, 3, 3)] * 8        # 48 tensors: two launches
    g = torch.Generator().manual_seed(5)
    init = [torch.randn(*s, generator=g) for s in shapes]
    mk = lambda: [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pa, pb = mk(), mk()
    groups = lambda ps: [{"params": ps[:20]}, {"params": ps[20:], "lr": 1e-3}]
    oa = RMSprop(groups(pa), lr=1e-2, weight_decay=5e-4)
    ob = torch.optim.RMSprop(groups(pb), lr=1e-2, weight_decay=5e-4)
    for it in range(3):
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).to(dev) * (it + 1)
            x.grad = gr.clone(); y.grad = gr.clone()
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        _close(x, y, 2e-6, "param")
    for x, y in zip(pa, pb):
        _close(oa.state[x]["square_avg"], ob.state[y]["square_avg"], 2e-6, "square_avg")
    ob.load_state_dict(oa.state_dict()); oa.load_state_dict(ob.state_dict())             # interchangeable layout


@pytest.mark.parametrize("seed", list(range(12)))
def test_conv_engines_random_shapes(dev, seed):
    """Seeded random conv geometries that reach the wide split-pipe tiles (>= 1024 output rows) with ragged everything:
    M and Cout tails, odd spatial sizes, stride 2, 1x1 and 3x3, 64-channel (256x64 tile) and 128+ channel layers —
    forward (with BatchNorm partial sums), data gradient and weight gradient against fp64."""
    import random as _r
    from dcnet_amd import ops
    rng = _r.Random(1000 + seed)
    k = rng.choice([1, 3]); st = rng.choice([1, 1, 2])
    cin = rng.choice([32, 64, 96, 128, 160, 256])
    cout = rng.choice([64, 96, 128, 136, 192, 256, 320])
    h = rng.randint(17, 45); w = rng.randint(17, 45); n = rng.randint(2, 4)
    x = _rand(n, h, w, cin, seed=seed).to(dev)
    wt = (_rand(cout, k, k, cin, seed=seed + 100) / (cin * k * k) ** 0.5).to(dev)
    xd = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wd = wt.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    yd = F.conv2d(xd, wd, stride=st, padding=(k - 1) // 2)
    dy = (_rand(*yd.shape, seed=seed + 200) / 8).permute(0, 2, 3, 1).contiguous().to(dev)
    yd.backward(dy.permute(0, 3, 1, 2).double().cpu())
    y, stats = ops.conv2d_fwd(x, wt, k, st, want_stats=True)
    _close(y, yd.detach().permute(0, 2, 3, 1), 3e-5, "fwd")
    ref_s = yd.detach().permute(0, 2, 3, 1).reshape(-1, cout)
    _close(stats[:, 0].double().sum(0), ref_s.sum(0), 1e-4, "stats sum")
    _close(stats[:, 1].double().sum(0), (ref_s * ref_s).sum(0), 1e-4, "stats sumsq")
    cout_p = (cout + 31) // 32 * 32
    wt_p = torch.zeros(cout_p, k, k, cin, device=dev); wt_p[:cout] = wt
    dy_p = torch.zeros(*dy.shape[:3], cout_p, device=dev); dy_p[..., :cout] = dy
    _close(ops.conv2d_bwd_data(dy_p, wt_p, (h, w), k, st), xd.grad.permute(0, 2, 3, 1), 3e-5, "dgrad")
    dw = ops.conv2d_bwd_weight(x, dy_p[..., :cout_p], k, st)[:cout] if cout % 4 else ops.conv2d_bwd_weight(x, dy, k, st)
    _close(dw, wd.grad.permute(0, 2, 3, 1), 3e-5, "wgrad")


STRIP_CASES = [
    # n, h, w, cin, cout     (3x3 stride 1; >= 1024 output rows so that the filter bank is pre-split)
    (8, 13, 13, 64, 128),      # 13x13 map, 128x128-or-256x128 tile, M tail
    (7, 13, 13, 32, 1024),     # many N tiles
    (3, 26, 26, 32, 64),       # 64 filters: stays on the implicit-GEMM tile (statistics partial per 256 rows)
    (2, 52, 52, 32, 128),
    (2, 40, 24, 96, 160),      # non-square map, ragged Cout
    (5, 17, 31, 64, 136),
    (1, 104, 104, 32, 128),    # wide map: strip of 466 pixels for 256 outputs
    (1, 208, 208, 32, 128),    # too wide for the strip (674 pixels): falls back
    (1, 33, 35, 32, 128),      # one image, odd sizes
]


@pytest.mark.parametrize("case", STRIP_CASES)
def test_conv3_strip_kernel(dev, case):
    """The 3x3 stride-1 strip kernel (csrc/conv3.hip: activations staged once per 16 channels for all nine taps) against
    fp64, and against the implicit-GEMM tile it replaces on the same inputs: forward with the fused epilogue, BatchNorm
    partial sums, shortcut add, a concat-slice destination, accumulate; data gradient plain and accumulating.  Both M tile
    sizes are forced.  Image borders, the wrap of the strip into neighbouring rows / images and the M / Cout tails all matter
    here."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout = case
    x = _rand(n, h, w, cin, seed=21).to(dev)
    wt = (_rand(cout, 3, 3, cin, seed=22) / (cin * 9) ** 0.5).to(dev)
    scale = (_rand(cout, seed=23).abs() + 0.5).to(dev); shift = _rand(cout, seed=24).to(dev)
    res = _rand(n, h, w, cout, seed=25).to(dev)
    xd = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wd = wt.permute(0, 3, 1, 2).double().cpu()
    raw = F.conv2d(xd, wd, padding=1)
    dy = (_rand(n, h, w, cout, seed=26) / 8)
    raw.backward(dy.permute(0, 3, 1, 2).double())
    rawl = raw.detach().permute(0, 2, 3, 1)
    ref = F.leaky_relu(rawl * scale.double().cpu() + shift.double().cpu(), 0.1) + res.double().cpu()
    cout_p = (cout + 31) // 32 * 32
    wt_p = torch.zeros(cout_p, 3, 3, cin, device=dev); wt_p[:cout] = wt
    dy_p = torch.zeros(n, h, w, cout_p, device=dev); dy_p[..., :cout] = dy.to(dev)
    base = _rand(n, h, w, cin, seed=27).to(dev)

    def run():
        out = {}
        out["y"], out["stats"] = ops.conv2d_fwd(x, wt, 3, 1, scale, shift, ops.ACT_LEAKY, 0.1, residual=res, want_stats=True)
        buf = torch.zeros(n, h, w, cout + 64, device=dev)
        am = ops.amax_slot(dev)
        ops.conv2d_fwd(x, wt, 3, 1, out=buf[..., 32:32 + cout], amax_out=am)
        out["slice"] = buf; out["amax"] = am.clone()
        acc = rawl.float().to(dev).clone()
        ops.conv2d_fwd(x, wt, 3, 1, out=acc, accumulate=True)
        out["acc"] = acc
        out["dx"] = ops.conv2d_bwd_data(dy_p, wt_p, (h, w), 3, 1)
        dx2 = base.clone()
        ops.conv2d_bwd_data(dy_p, wt_p, (h, w), 3, 1, out=dx2, accumulate=True)
        out["dx_acc"] = dx2
        return out

    try:
        lib().set_tuning(b"3x3strip", 0)
        old = run()
        ran16 = False
        # (bm, m16): both M tiles of conv3.hip forced (m16 = 0: 32x32x16 MFMAs), then conv3x.hip (16x16x32 MFMAs, two taps per MFMA) where the
        # 256-row tile is the choice (1, the default) and for every launch it fits (2)
        for bm, m16 in ((0, 0), (128, 0), (256, 0), (0, 1), (0, 2)):
            lib().set_tuning(b"3x3strip", 1); lib().set_tuning(b"3bm", bm); lib().set_tuning(b"3m16", m16)
            new = run()
            if m16 == 0:
                new32 = new
            elif m16 == 2:
                ran16 = not torch.equal(new["y"], new32["y"]) or not torch.equal(new["dx"], new32["dx"])
            _close(new["y"], ref, 2e-5, f"strip fwd epilogue bm={bm} m16={m16}")
            _close(new["stats"][:, 0].double().sum(0), rawl.reshape(-1, cout).sum(0), 1e-4, "strip stats sum")
            _close(new["stats"][:, 1].double().sum(0), (rawl * rawl).reshape(-1, cout).sum(0), 1e-4, "strip stats sumsq")
            assert new["stats"].shape == old["stats"].shape
            _close(new["slice"][..., 32:32 + cout], rawl, 2e-5, "strip slice")
            assert float(new["slice"][..., :32].abs().max()) == 0 and float(new["slice"][..., 32 + cout:].abs().max()) == 0
            assert float(new["amax"].view(torch.float32).max()) == float(new["slice"].abs().max())
            _close(new["acc"], 2 * rawl, 2e-5, "strip accumulate")
            _close(new["dx"], xd.grad.permute(0, 2, 3, 1), 2e-5, "strip dgrad")
            _close(new["dx_acc"], xd.grad.permute(0, 2, 3, 1) + base.double().cpu(), 2e-5, "strip dgrad accumulate")
            # same split arithmetic, different summation order over the taps only: agreement far inside the fp64 tolerance
            for k in ("y", "dx"):
                _close(new[k], old[k], 2e-6, f"strip vs tile {k}")
        if cout >= 128 and w <= 64:         # (64-filter layers and maps wider than ~120 pixels stay on the implicit-GEMM tile)
            assert not torch.equal(new["y"], old["y"]) or not torch.equal(new["dx"], old["dx"]), "the strip kernel did not run"
        if cout >= 128 and cout % 4 == 0 and w <= 104:
            assert ran16, "conv3x.hip did not run"
    finally:
        lib().set_tuning(b"3x3strip", 1); lib().set_tuning(b"3bm", 0); lib().set_tuning(b"3m16", 1)


@pytest.mark.parametrize("case", [(2, 26, 26, 512, 256, 1, 1), (8, 13, 13, 1024, 512, 1, 1), (1, 37, 29, 32, 128, 1, 1), (2, 52, 52, 256, 128, 1, 1),
                                  (1, 33, 32, 96, 384, 1, 1), (8, 13, 13, 64, 128, 1, 1), (2, 40, 36, 64, 32, 1, 1), (3, 30, 26, 128, 64, 1, 1),
                                  (2, 52, 50, 32, 64, 3, 2), (2, 45, 47, 64, 128, 3, 2), (1, 64, 60, 32, 64, 3, 1), (2, 37, 41, 64, 64, 3, 1),
                                  (8, 26, 26, 128, 256, 3, 2)])
def test_conv1_lds_dma_kernel(dev, case):
    """The implicit-GEMM kernel with both tiles by LDS-DMA (csrc/conv1.hip: rings of K-steps in flight across raw barriers, the f16
    split done on the fragment) against fp64 and against the implicit-GEMM tiles it replaces: 1x1 layers (plain GEMM rows), 3x3
    stride-2 layers and the four parity classes of their data gradients, 3x3 stride-1 layers with 32 / 64 filters; forward with the
    fused epilogue, BatchNorm partial sums, shortcut add, concat-slice destination, accumulate, abs-max word; data gradient plain
    and accumulating.  Several ring depths; K loops shorter than the ring, M tails, odd image sizes (borders), every tile shape."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout, k, st = case
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
    x = _rand(n, h, w, cin, seed=41).to(dev)
    wt = (_rand(cout, k, k, cin, seed=42) / (cin * k * k) ** 0.5).to(dev)
    scale = (_rand(cout, seed=43).abs() + 0.5).to(dev); shift = _rand(cout, seed=44).to(dev)
    res = _rand(n, ho, wo, cout, seed=45).to(dev)
    xd = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    raw = F.conv2d(xd, wt.permute(0, 3, 1, 2).double().cpu(), stride=st, padding=pad)
    dy = _rand(n, ho, wo, cout, seed=46) / 8
    raw.backward(dy.permute(0, 3, 1, 2).double())
    rawl = raw.detach().permute(0, 2, 3, 1)
    dxref = xd.grad.permute(0, 2, 3, 1)
    ref = F.leaky_relu(rawl * scale.double().cpu() + shift.double().cpu(), 0.1) + res.double().cpu()
    dyd = dy.to(dev)
    base = _rand(n, h, w, cin, seed=47).to(dev)

    def run():
        out = {}
        out["y"], out["stats"] = ops.conv2d_fwd(x, wt, k, st, scale, shift, ops.ACT_LEAKY, 0.1, residual=res, want_stats=True)
        buf = torch.zeros(n, ho, wo, cout + 64, device=dev)
        am = ops.amax_slot(dev)
        ops.conv2d_fwd(x, wt, k, st, out=buf[..., 32:32 + cout], amax_out=am)
        out["slice"] = buf; out["amax"] = am.clone()
        acc = rawl.float().to(dev).clone()
        _, out["acc_stats"] = ops.conv2d_fwd(x, wt, k, st, out=acc, accumulate=True, want_stats=True)
        out["acc"] = acc
        out["dx"] = ops.conv2d_bwd_data(dyd, wt, (h, w), k, st)
        dx2 = base.clone()
        ops.conv2d_bwd_data(dyd, wt, (h, w), k, st, out=dx2, accumulate=True)
        out["dx_acc"] = dx2
        return out

    import ctypes

    def launches():
        lib().prof_enable(1)
        run()
        lib().prof_enable(0)
        c = (ctypes.c_int64 * 64)(); m = (ctypes.c_double * 64)(); wk = (ctypes.c_double * 64)()
        lib().prof_collect(ctypes.addressof(c), ctypes.addressof(m), ctypes.addressof(wk), 0)
        return c[35]

    try:
        lib().set_tuning(b"1x1dma", 0)
        old = run()
        assert launches() == 0
        lib().set_tuning(b"1x1dma", 1)
        for stages in (32, 33, 44, 63):
            lib().set_tuning(b"1stages", stages)
            new = run()
            _close(new["y"], ref, 2e-5, f"conv1 fwd epilogue stages={stages}")
            _close(new["stats"][:, 0].double().sum(0), rawl.reshape(-1, cout).sum(0), 1e-4, "conv1 stats sum")
            _close(new["stats"][:, 1].double().sum(0), (rawl * rawl).reshape(-1, cout).sum(0), 1e-4, "conv1 stats sumsq")
            assert new["stats"].shape == old["stats"].shape
            _close(new["slice"][..., 32:32 + cout], rawl, 2e-5, "conv1 slice")
            assert float(new["slice"][..., :32].abs().max()) == 0 and float(new["slice"][..., 32 + cout:].abs().max()) == 0
            assert float(new["amax"].view(torch.float32).max()) == float(new["slice"].abs().max())
            _close(new["acc"], 2 * rawl, 2e-5, "conv1 accumulate")
            _close(new["acc_stats"][:, 0].double().sum(0), 2 * rawl.reshape(-1, cout).sum(0), 1e-4, "conv1 accumulate stats (raw + content)")
            _close(new["dx"], dxref, 2e-5, "conv1 dgrad")
            _close(new["dx_acc"], dxref + base.double().cpu(), 2e-5, "conv1 dgrad accumulate")
            for kk in ("y", "dx"):        # the same split arithmetic (another summation order over taps at most)
                _close(new[kk], old[kk], 2e-6, f"conv1 vs tile {kk}")
        # (for plain GEMM rows both kernels add the three cross terms of every 16-deep K-step in the same order: bitwise equal
        #  results, so "it ran" is read off the library's launch records: tag 35 = conv1_kernel)
        if cout % 128 == 0 and (k == 1 or st == 2):
            assert launches() >= 3, "the LDS-DMA kernel did not take the forward launches"
        # (32 / 64 filters on the fp32-pipe narrow tiles and 3x3 stride-1 launches stay where they were: csrc/conv1.hip conv1_will_take)
    finally:
        lib().set_tuning(b"1x1dma", 1); lib().set_tuning(b"1stages", 32)


@pytest.mark.parametrize("case", [(8, 13, 13, 64, 128), (2, 52, 52, 32, 128), (2, 40, 24, 96, 160), (5, 17, 31, 64, 136),
                                  (1, 104, 104, 32, 128)])
def test_conv3_strip_kernel_bf16_operands(dev, case):
    """bf16-operand mode (configs[2]) on the strip kernel: one bf16 plane per operand, the filter bank in bf16 from
    ops.FilterBanks.  Against the exact model of the mode — fp64 convolution of the bf16-rounded tensors — forward with the
    fused epilogue and BatchNorm sums, data gradient; and against the implicit-GEMM tile of the same mode."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout = case
    cout_p = (cout + 31) // 32 * 32
    x = _rand(n, h, w, cin, seed=31).to(dev)
    w_oihw = torch.zeros(cout_p, cin, 3, 3)
    w_oihw[:cout] = _rand(cout, cin, 3, 3, seed=32) / (cin * 9) ** 0.5
    w_oihw = w_oihw.to(dev)
    scale = (_rand(cout_p, seed=33).abs() + 0.5).to(dev); shift = _rand(cout_p, seed=34).to(dev)
    dy = torch.zeros(n, h, w, cout_p); dy[..., :cout] = _rand(n, h, w, cout, seed=36) / 8
    dy = dy.to(dev)
    rb = lambda t: t.to(torch.bfloat16).double().cpu()
    xd = rb(x).permute(0, 3, 1, 2).requires_grad_(True)
    raw = F.conv2d(xd, rb(w_oihw), padding=1)
    raw.backward(rb(dy).permute(0, 3, 1, 2))
    rawl = raw.detach().permute(0, 2, 3, 1)
    ref = F.leaky_relu(rawl * scale.double().cpu() + shift.double().cpu(), 0.1)
    fb = ops.FilterBanks({0: w_oihw}, dev)
    fb.refresh()
    b = fb.get(0, w_oihw)
    assert torch.equal(b["b16"].view(cout_p, 3, 3, cin), w_oihw.permute(0, 2, 3, 1).to(torch.bfloat16))
    assert torch.equal(b["tb16"].view(cin, 3, 3, cout_p), w_oihw.permute(1, 2, 3, 0).to(torch.bfloat16))
    try:
        ops.set_precision("bf16")
        lib().set_tuning(b"3x3strip", 0)
        y0, _ = ops.conv2d_fwd(x, b["ohwi"], 3, 1, scale, shift, ops.ACT_LEAKY, 0.1, want_stats=True, w_b16=b["b16"])
        d0 = ops.conv2d_bwd_data(dy, b["ohwi"], (h, w), 3, 1, wt_ready=(b["t"], b["tsplit"]), wt_b16=b["tb16"])
        lib().set_tuning(b"3x3strip", 1)
        for bm in (0, 128, 256):
            lib().set_tuning(b"3bm", bm)
            y1, st = ops.conv2d_fwd(x, b["ohwi"], 3, 1, scale, shift, ops.ACT_LEAKY, 0.1, want_stats=True, w_b16=b["b16"])
            d1 = ops.conv2d_bwd_data(dy, b["ohwi"], (h, w), 3, 1, wt_ready=(b["t"], b["tsplit"]), wt_b16=b["tb16"])
            _close(y1, ref, 2e-5, f"bf16 strip fwd bm={bm}")
            _close(st[:, 0].double().sum(0), rawl.reshape(-1, cout_p).sum(0), 1e-4, "bf16 strip stats")
            # (a data gradient with <= 64 input channels runs on the narrow fp32-operand tiles in every mode: closer to fp32 than
            #  to the bf16 model)
            _close(d1, xd.grad.permute(0, 2, 3, 1), 2e-5 if cin > 64 else 1e-2, f"bf16 strip dgrad bm={bm}")
            _close(y1, y0, 2e-6, "bf16 strip vs tile fwd"); _close(d1, d0, 2e-6, "bf16 strip vs tile dgrad")
        if w <= 64:
            assert not torch.equal(y1, y0) or not torch.equal(d1, d0), "the strip kernel did not run"
        # without the bf16 bank the mode stays on the implicit-GEMM tile: same exact model
        y2, _ = ops.conv2d_fwd(x, b["ohwi"], 3, 1, scale, shift, ops.ACT_LEAKY, 0.1)
        assert torch.equal(y2, y0)
    finally:
        lib().set_tuning(b"3x3strip", 1); lib().set_tuning(b"3bm", 0)
        ops.set_precision("fp32")
    # it IS reduced precision: the fp32-mode result differs from the bf16 model by far more than the tolerance above
    yf, _ = ops.conv2d_fwd(x, b["ohwi"], 3, 1, scale, shift, ops.ACT_LEAKY, 0.1)
    assert float((yf.double().cpu() - ref).abs().max()) > 1e-4


def test_filter_banks_match_the_per_layer_preparation(dev):
    """ops.FilterBanks (dcn_prepare_filters: every filter bank of a network in three launches) writes the same OHWI bank,
    transposed bank, abs-max and f16-split banks as the per-layer kernels it replaces, and convolutions fed from it give
    bitwise the results of the per-layer path (forward and data gradient, 3x3 and 1x1, strides 1 and 2)."""
    from dcnet_amd import ops
    shapes = [(128, 64, 3, 3), (64, 128, 1, 1), (256, 96, 3, 3), (96, 32, 1, 1), (33, 64, 3, 3), (64, 3, 3, 3)]
    ws = {i: (_rand(*sh, seed=70 + i) * (10.0 ** (i - 2))).to(dev) for i, sh in enumerate(shapes)}
    fb = ops.FilterBanks(ws, dev)
    assert fb.valid_for(ws) and fb.get(4, ws[4]) is None and fb.get(5, ws[5]) is None      # 33 filters / 3 channels: per-layer path
    fb.refresh()
    for i in range(4):
        w = ws[i]; co, ci, kh, kw = w.shape
        b = fb.get(i, w)
        assert torch.equal(b["ohwi"], w.permute(0, 2, 3, 1).contiguous())
        assert torch.equal(b["t"][:w.numel()].view(ci, kh, kw, co), w.permute(1, 2, 3, 0).contiguous())
        assert float(b["amax"].view(torch.float32).max()) == float(w.abs().max())
        for name, ref in (("split", w.permute(0, 2, 3, 1)), ("tsplit", w.permute(1, 2, 3, 0))):
            s = float(b[name][w.numel()])
            hl = b[name][:w.numel()].view(-1, 8).view(torch.float16).view(-1, 2, 8).double()
            back = (hl[:, 0] + hl[:, 1]).reshape(-1) / s
            amax = float(w.abs().max())
            assert 2.0 ** 13 <= amax * s < 2.0 ** 14
            assert float((back - ref.reshape(-1).double()).abs().max()) <= amax * 2.0 ** -21
    x = _rand(4, 20, 20, 64, seed=80).to(dev)
    for i, k, st in ((0, 3, 1), (0, 3, 2)):
        b = fb.get(i, ws[i])
        y0, _ = ops.conv2d_fwd(x, b["ohwi"], k, st)
        y1, _ = ops.conv2d_fwd(x, b["ohwi"], k, st, amax_w=b["amax"], w_split_ready=b["split"])
        assert torch.equal(y0, y1)
        dy = _rand(*y0.shape, seed=81).to(dev)
        d0 = ops.conv2d_bwd_data(dy, b["ohwi"], (20, 20), k, st)
        d1 = ops.conv2d_bwd_data(dy, b["ohwi"], (20, 20), k, st, amax_w=b["amax"], wt_ready=(b["t"], b["tsplit"]))
        assert torch.equal(d0, d1)
    x1 = _rand(4, 20, 20, 128, seed=82).to(dev)
    b = fb.get(1, ws[1])
    assert torch.equal(ops.conv2d_fwd(x1, b["ohwi"], 1, 1)[0], ops.conv2d_fwd(x1, b["ohwi"], 1, 1, amax_w=b["amax"], w_split_ready=b["split"])[0])


D2_CASES = [
    # n, h, w, cin   (cin -> 2 cin 3x3 stride 2: dY (n, h/2, w/2, 2 cin) -> dX (n, h, w, cin); >= 65536 input pixels, output rows of >= 64)
    (4, 128, 128, 32),
    (2, 130, 256, 32),
    (3, 132, 176, 32),      # 89 padded entries per row: a chunk of 64 wraps in most steps, images wrap too
    (1, 264, 256, 32),
    (4, 128, 128, 64),
    (3, 132, 176, 64),
    (1, 264, 258, 64),
]


@pytest.mark.parametrize("case", D2_CASES)
def test_dgrad2_register_bank_kernel(dev, case):
    """csrc/nconv.hip (data gradients of the 32 -> 64 and 64 -> 128 stride-2 layers: four parity classes from one pass over dY,
    filter bank in registers, persistent workgroups) against fp64 and against the implicit-GEMM launch it replaces: image borders
    (the row below the last dY row and the column right of the last column read zeros), chunk / row / image wraps, a dY that is a
    slice of a wider tensor, accumulation into an existing gradient, with and without the prepared banks."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin = case
    cout = 2 * cin
    wgt = (_rand(cout, cin, 3, 3, seed=61) / 6)
    wide = _rand(n, h // 2, w // 2, cout + 32, seed=62).to(dev)
    dy = wide[..., 16:16 + cout]                                   # pixel stride cout + 32
    xd = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, wgt.double(), padding=1, stride=2).backward(dy.permute(0, 3, 1, 2).double().cpu())
    ref = xd.grad.permute(0, 2, 3, 1)
    w_ohwi = wgt.permute(0, 2, 3, 1).contiguous().to(dev)
    wdev = wgt.to(dev)
    fb = ops.FilterBanks([wdev], dev); fb.refresh()
    b = fb.get(0, wdev)
    base = _rand(n, h, w, cin, seed=63).to(dev)
    try:
        lib().set_tuning(b"Nconv", 0)
        old = ops.conv2d_bwd_data(dy, w_ohwi, (h, w), 3, 2)
        lib().set_tuning(b"Nconv", 1)
        lib().prof_enable(1)
        new = ops.conv2d_bwd_data(dy, w_ohwi, (h, w), 3, 2)
        lib().prof_enable(0)
        ran = _prof_launches(37)
        new_c = ops.conv2d_bwd_data(dy.contiguous(), w_ohwi, (h, w), 3, 2)
        new_b = ops.conv2d_bwd_data(dy, b["ohwi"], (h, w), 3, 2, amax_w=b["amax"], wt_ready=(b["t"], b["tsplit"]))
        acc = base.clone()
        ops.conv2d_bwd_data(dy, w_ohwi, (h, w), 3, 2, out=acc, accumulate=True)
    finally:
        lib().set_tuning(b"Nconv", 1); lib().prof_enable(0)
    assert ran == 1, "the register-bank kernel did not run"
    _close(new, ref, 2e-5, "dgrad2")
    _close(old, ref, 2e-5, "implicit-GEMM classes")
    _close(acc, ref + base.double().cpu(), 2e-5, "dgrad2 accumulate")
    # BatchNorm tap: the partial sums of the backward of the layer in front (y_prev -> bn -> LeakyReLU -> this convolution), formed in
    # the epilogue, against the reduce pass over (y_prev, dx)
    y_prev = _rand(n, h, w, cin, seed=64).to(dev)
    mean = (_rand(cin, seed=65) / 4).to(dev); invstd = (torch.rand(cin, generator=torch.Generator().manual_seed(66)) + 0.5).to(dev)
    gamma = (torch.rand(cin, generator=torch.Generator().manual_seed(67)) + 0.5).to(dev); beta = (_rand(cin, seed=68) / 4).to(dev)
    tap = dict(y=y_prev, mean=mean, invstd=invstd, gamma=gamma, beta=beta, act=ops.ACT_LEAKY, slope=0.1)
    dx_t, part = ops.conv2d_bwd_data(dy, w_ohwi, (h, w), 3, 2, tap=tap)
    assert torch.equal(dx_t, new)
    if cin == 32:
        assert part is not None and part.shape[1:] == (2, cin)
        want = ops._bn_bwd_partials(y_prev, new, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1, None)
        want = want[0][:want[1] * 2 * cin].view(want[1], 2, cin).double().sum(0)
        _close(part.double().sum(0), want, 2e-5, "tap partial sums")
        dy1, dg1, db1 = ops.bn_act_bwd(y_prev, new, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1)
        dy2, dg2, db2 = ops.bn_act_bwd(y_prev, new, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1, part=part)
        _close(dg2, dg1, 2e-5, "dgamma from the tap"); _close(db2, db1, 2e-5, "dbeta from the tap"); _close(dy2, dy1, 2e-5, "dy from the tap")
    else:
        assert part is None                                        # (the 64-channel form has no registers left for it)
    assert torch.equal(new, new_c)                                 # the pixel stride of dY changes nothing
    assert torch.equal(new, new_b)                                 # nor do the prepared banks
    assert torch.equal(new, ops.conv2d_bwd_data(dy, w_ohwi, (h, w), 3, 2))


N1_CASES = [
    # mode, stride, n, h, w   (input size; 32 <-> 64 channels, 3x3; >= 65536 output pixels, output rows of >= 64)
    ("fwd", 1, 4, 128, 128),
    ("fwd", 1, 3, 132, 176),
    ("fwd", 1, 1, 258, 255),     # odd width: 256 padded entries per row
    ("fwd", 2, 4, 256, 256),
    ("fwd", 2, 3, 264, 352),
    ("fwd", 2, 2, 520, 260),
    ("dgrad", 1, 4, 128, 128),
    ("dgrad", 1, 3, 132, 176),
    ("dgrad", 1, 1, 258, 255),
]


PRE_CASES = [
    # stride, n, h, w   (32 -> 64 channels, 3x3; >= 65536 output pixels, output rows of >= 64)
    (1, 4, 128, 128),
    (1, 1, 258, 255),
    (2, 4, 256, 256),
    (2, 3, 264, 352),
    (2, 2, 520, 260),
]


G3_CASES = [
    # batch, M, N, K, form   (forms: "nt" A [M][K] B [N][K]; "nn" B [K][N]; "tn" A [K][M], B [K][N])
    (2, 512, 512, 64, "nt"),
    (3, 700, 530, 96, "nt"),       # ragged M / N tiles
    (2, 2704, 2704, 512, "nt"),    # the affinity of the 52 x 52 maps
    (2, 676, 256, 676, "nn"),      # K = 676: a 4-wide last slice; E rows padded to 704
    (3, 600, 512, 1000, "nn"),
    (2, 2704, 512, 2704, "nn"),
    (2, 676, 256, 676, "tn"),
    (3, 600, 512, 1000, "tn"),
    (2, 2704, 512, 2704, "tn"),
]


@pytest.mark.parametrize("case", G3_CASES)
def test_gemm3_presplit_operands(dev, case):
    """csrc/gemm3.hip (batched GEMM on pre-split operands: both tiles by LDS-DMA, K along the row or across rows from the same bytes)
    against fp64 in its three forms, with ragged tiles, padded rows (pads must be zero, rows / k beyond the tensors must read as
    zeros), strided batches, the row scale and accumulation."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    bsz, M, N, K, form = case
    assert lib().gemm3_supported(M, N, K, bsz)
    pad32 = lambda v: (v + 31) // 32 * 32
    a_t, b_t = form == "tn", form != "nt"
    a_shape = (K, M) if a_t else (M, K)
    b_shape = (K, N) if b_t else (N, K)
    # operands as views of wider, batch-interleaved buffers (row stride > columns; pads zero as the contract asks)
    def make(shape, seed, scale):
        rows, cols = shape
        buf = torch.zeros(bsz, 2, rows, pad32(cols) + 32, device=dev)
        v = buf[:, 1, :, :pad32(cols)]
        v[:, :, :cols] = (_rand(bsz, rows, cols, seed=seed) * scale).to(dev)
        return v, cols
    a, a_cols = make(a_shape, 81, 3.0)
    b, b_cols = make(b_shape, 82, 0.05)
    ad, bd = a[:, :, :a_cols].double().cpu(), b[:, :, :b_cols].double().cpu()
    A2 = ad.transpose(1, 2) if a_t else ad          # (b, M, K)
    B2 = bd if b_t else bd.transpose(1, 2)          # (b, K, N)
    ref = torch.bmm(A2, B2)
    am_a, am_b = ops.absmax(a.contiguous()), ops.absmax(b.contiguous())
    a_s = ops.gemm3_presplit(a, am_a)
    b_s = a_s.new_zeros(b.shape); ops.gemm3_presplit(b, am_b, out=b_s)
    # in place gives the same bytes
    a_ip = a.clone(); ops.gemm3_presplit(a_ip, am_a, out=a_ip)
    assert torch.equal(a_ip.view(torch.int32), a_s.view(torch.int32))
    out = torch.full((bsz, M, N + 32), 7.0, device=dev)
    c = out[:, :, 16:16 + N]
    try:
        lib().prof_enable(1)
        ops.gemm3(a_s, b_s, c, M, N, K, am_a, am_b, a_t=a_t, b_t=b_t)
        lib().prof_enable(0)
        ran = _prof_launches(40)
    finally:
        lib().prof_enable(0)
    assert ran == 1
    _close(c, ref, 2e-5, f"gemm3 {form}")
    assert float((out[:, :, :16] - 7).abs().max()) == 0 and float((out[:, :, 16 + N:] - 7).abs().max()) == 0
    rs = (torch.rand(bsz, M, generator=torch.Generator().manual_seed(83)) + 0.5).to(dev)
    base = c.clone()
    ops.gemm3(a_s, b_s, c, M, N, K, am_a, am_b, a_t=a_t, b_t=b_t, row_scale=rs, accumulate=True)
    _close(c, ref * (1 + rs.double().cpu().unsqueeze(2)), 3e-5, f"gemm3 {form} row scale + accumulate")
    again = torch.full_like(out, 7.0)[:, :, 16:16 + N]
    ops.gemm3(a_s, b_s, again, M, N, K, am_a, am_b, a_t=a_t, b_t=b_t)
    assert torch.equal(again, base)


TAP1_CASES = [
    # n, h, w, cin, cout, k   (stride 1; the data gradient goes cout -> cin channels and taps the BatchNorm of the cin-channel layer in front)
    (4, 26, 26, 256, 128, 1),      # conv1.hip, accumulating into a shortcut gradient
    (2, 52, 52, 256, 128, 1),
    (8, 13, 13, 1024, 512, 1),
    (4, 26, 26, 128, 256, 3),      # conv3.hip
    (2, 52, 52, 128, 256, 3),
    (8, 13, 21, 256, 512, 3),      # ragged last M-tile
]


@pytest.mark.parametrize("m16", [0, 2])
@pytest.mark.parametrize("case", TAP1_CASES)
def test_bn_tap_on_stride1_data_gradients(dev, case, m16):
    """The BatchNorm tap in the partial-sum epilogues of csrc/conv1.hip / conv3.hip: the data gradient of a stride-1 layer forms
    sum(g) and sum(g * xhat) of the BatchNorm + LeakyReLU in front (g = dx * act') while dx is in registers — against
    dcn_bn_act_bwd_reduce over the written dx, with and without accumulation into an existing gradient."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout, k = case
    if m16 and k != 3:
        pytest.skip("conv3x.hip only takes 3x3 launches")
    lib().set_tuning(b"3m16", m16)           # conv3.hip (0) / conv3x.hip wherever it fits (2)
    wgt = (_rand(cout, cin, k, k, seed=101) / (3 * k)).to(dev)
    w_ohwi = wgt.permute(0, 2, 3, 1).contiguous()
    dy = _rand(n, h, w, cout, seed=102).to(dev)
    y_prev = (_rand(n, h, w, cin, seed=103) * 2).to(dev)
    gamma = (_rand(cin, seed=104)).to(dev); beta = (_rand(cin, seed=105) / 2).to(dev)      # (negative gammas too)
    mean = y_prev.reshape(-1, cin).mean(0); invstd = 1.0 / torch.sqrt(y_prev.reshape(-1, cin).var(0, unbiased=False) + 1e-5)
    tap = dict(y=y_prev, mean=mean, invstd=invstd, gamma=gamma, beta=beta, act=ops.ACT_LEAKY, slope=0.1)
    for accumulate in (False, True):
        base = _rand(n, h, w, cin, seed=106).to(dev)
        plain = base.clone() if accumulate else None
        plain = ops.conv2d_bwd_data(dy, w_ohwi, (h, w), k, 1, out=plain, accumulate=accumulate)
        tapped = base.clone() if accumulate else None
        tapped, part = ops.conv2d_bwd_data(dy, w_ohwi, (h, w), k, 1, out=tapped, accumulate=accumulate, tap=tap)
        assert part is not None, "the launch did not tap"
        assert torch.equal(plain, tapped)
        ref, r_ = ops._bn_bwd_partials(y_prev, plain, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1, None)
        ref = ref[:r_ * 2 * cin].view(r_, 2, cin).clone()
        _close(part.double().sum(0), ref.double().sum(0), 2e-5, f"tap partial sums (accumulate={accumulate})")
        # the BatchNorm backward fed with the tapped partials equals the one that reduces by itself
        d0 = ops.bn_act_bwd(y_prev, plain, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1)
        d1 = ops.bn_act_bwd(y_prev, plain, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1, part=part)
        for a_, b_ in zip(d0, d1):
            _close(a_, b_, 2e-5, "bn_act_bwd with tapped partials")
    lib().set_tuning(b"3m16", 1)


@pytest.mark.parametrize("case", PRE_CASES)
def test_loader_side_activation(dev, case):
    """dcn_conv2d_fwd_pre / dcn_conv2d_bwd_weight_pre (the BatchNorm scale / shift + LeakyReLU of the layer in front applied where the
    input is staged, csrc/nconv.hip and csrc/wgrad9.hip) against dcn_scale_act followed by the plain calls, and against fp64.  The shifts
    are large so that a pad read as act(shift) instead of zero would show at every border."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    st, n, h, w = case
    assert ops.pre_supported(n, h, w, 32, 64, 3, st)
    assert not ops.pre_supported(n, h, w, 64, 128, 3, st) and not ops.pre_supported(2, 16, 16, 32, 64, 3, st)
    y_raw = (_rand(n, h, w, 32, seed=61) * 3).to(dev)
    scale = (_rand(32, seed=62) * 1.5).to(dev); shift = (_rand(32, seed=63) * 2 + 1).to(dev)
    wgt = (_rand(64, 32, 3, 3, seed=64) / 6)
    w_ohwi = wgt.permute(0, 2, 3, 1).contiguous().to(dev)
    aw = ops.absmax(w_ohwi)
    ay = ops.absmax(y_raw)
    act = ops.scale_act(y_raw, scale, shift, ops.ACT_LEAKY, 0.1)
    bound = ops.bn_act_amax_bound(ay, scale, shift, 0.1)
    b_, a_ = bound.view(torch.float32), ops.absmax(act).view(torch.float32)
    assert float(b_.min()) == float(b_.max()) >= float(a_.max()) > 0
    want = float((scale.abs() * float(y_raw.abs().max()) + shift.abs()).max())
    assert abs(float(b_.max()) - want) <= 1e-6 * want
    pre = ops.PreAct(y_raw, scale, shift, ops.ACT_LEAKY, 0.1)
    dy = (_rand(n, h // st, w // st, 96, seed=65) / 8).to(dev)[..., 16:80]
    # fp64 truth from the fp32 activation (the operand both paths see, bit for bit)
    ad = act.permute(0, 3, 1, 2).double().cpu()
    wd_ = wgt.double().requires_grad_(True)
    ref = F.conv2d(ad, wd_, padding=1, stride=st)
    ref.backward(dy.permute(0, 3, 1, 2).double().cpu())
    ref = ref.detach().permute(0, 2, 3, 1); ref_dw = wd_.grad.permute(0, 2, 3, 1)
    try:
        lib().prof_enable(1)
        y1, s1 = ops.conv2d_fwd(pre, w_ohwi, 3, st, want_stats=True, amax_x=bound, amax_w=aw)
        dw1 = ops.conv2d_bwd_weight(pre, dy, 3, st, amax_x=bound)
        lib().prof_enable(0)
        ran = (_prof_launches(38), _prof_launches(36))
    finally:
        lib().prof_enable(0)
    assert ran == (1, 1), f"the register-bank kernels did not run: {ran}"
    y0, s0 = ops.conv2d_fwd(act, w_ohwi, 3, st, want_stats=True, amax_w=aw)
    dw0 = ops.conv2d_bwd_weight(act, dy, 3, st)
    _close(y1, ref, 2e-5, "conv2d_fwd_pre"); _close(y0, ref, 2e-5, "scale_act + conv2d_fwd")
    _close(dw1, ref_dw, 3e-5, "conv2d_bwd_weight_pre"); _close(dw0, ref_dw, 3e-5, "scale_act + conv2d_bwd_weight")
    rawl = ref.reshape(-1, 64)
    _close(s1[:, 0].double().sum(0), rawl.sum(0), 1e-4, "pre stats sum")
    _close(s1[:, 1].double().sum(0), (rawl * rawl).sum(0), 1e-4, "pre stats sumsq")
    # with the SAME abs-max word the operand pieces are the same bits: identical results
    y2, s2 = ops.conv2d_fwd(act, w_ohwi, 3, st, want_stats=True, amax_x=bound, amax_w=aw)
    dw2 = ops.conv2d_bwd_weight(act, dy, 3, st, amax_x=bound)
    assert torch.equal(y1, y2) and torch.equal(s1, s2) and torch.equal(dw1, dw2)
    # no activation: the loader only scales and shifts
    lin = ops.PreAct(y_raw, scale, shift, ops.ACT_NONE, 0.0)
    act_l = ops.scale_act(y_raw, scale, shift, ops.ACT_NONE, 0.0)
    bl = ops.bn_act_amax_bound(ay, scale, shift, 0.0)
    assert torch.equal(ops.conv2d_fwd(lin, w_ohwi, 3, st, amax_x=bl, amax_w=aw)[0], ops.conv2d_fwd(act_l, w_ohwi, 3, st, amax_x=bl, amax_w=aw)[0])
    assert torch.equal(ops.conv2d_bwd_weight(lin, dy, 3, st, amax_x=bl), ops.conv2d_bwd_weight(act_l, dy, 3, st, amax_x=bl))
    with pytest.raises(Exception):
        ops.conv2d_fwd(pre, w_ohwi, 3, st, scale=scale, amax_x=bound, amax_w=aw)


@pytest.mark.parametrize("case", N1_CASES)
def test_nconv1_register_bank_kernels(dev, case):
    """csrc/nconv.hip nconv1_kernel (32 -> 64 3x3 forward at stride 1 and 2 with BatchNorm partial sums, and the stride-1 data
    gradient 64 -> 32; filter bank in registers, persistent workgroups over padded output positions) against fp64 and against the
    implicit-GEMM tiles they replace: image borders, chunk / row / image wraps, the stride-2 even / odd planes, sliced outputs."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    mode, st, n, h, w = case
    wgt = (_rand(64, 32, 3, 3, seed=71) / 6)
    w_ohwi = wgt.permute(0, 2, 3, 1).contiguous().to(dev)
    if mode == "fwd":
        x = _rand(n, h, w, 32, seed=72).to(dev)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wgt.double(), padding=1, stride=st).permute(0, 2, 3, 1)
        ho, wo = ref.shape[1], ref.shape[2]

        def run():
            y, stats = ops.conv2d_fwd(x, w_ohwi, 3, st, want_stats=True)
            buf = torch.zeros(n, ho, wo, 128, device=dev)
            ops.conv2d_fwd(x, w_ohwi, 3, st, out=buf[..., 32:96])
            return y, stats, buf
    else:
        wide = _rand(n, h, w, 96, seed=73).to(dev)
        dy = wide[..., 16:80]
        xd = torch.zeros(n, 32, h, w, dtype=torch.float64, requires_grad=True)
        F.conv2d(xd, wgt.double(), padding=1).backward(dy.permute(0, 3, 1, 2).double().cpu())
        ref = xd.grad.permute(0, 2, 3, 1)

        def run():
            return ops.conv2d_bwd_data(dy, w_ohwi, (h, w), 3, 1), None, ops.conv2d_bwd_data(dy.contiguous(), w_ohwi, (h, w), 3, 1)
    try:
        lib().set_tuning(b"Nconv", 0)
        old = run()
        lib().set_tuning(b"Nconv", 1)
        lib().prof_enable(1)
        new = run()
        lib().prof_enable(0)
        ran = _prof_launches(38)
    finally:
        lib().set_tuning(b"Nconv", 1); lib().prof_enable(0)
    assert ran == 2, "the register-bank kernel did not run"
    _close(new[0], ref, 2e-5, f"nconv1 {mode}")
    _close(old[0], ref, 2e-5, "implicit-GEMM tile")
    if mode == "fwd":
        rawl = ref.reshape(-1, 64)
        _close(new[1][:, 0].double().sum(0), rawl.sum(0), 1e-4, "nconv1 stats sum")
        _close(new[1][:, 1].double().sum(0), (rawl * rawl).sum(0), 1e-4, "nconv1 stats sumsq")
        assert new[1].shape == old[1].shape
        assert torch.equal(new[2][..., 32:96], new[0])
        assert float(new[2][..., :32].abs().max()) == 0 and float(new[2][..., 96:].abs().max()) == 0
    else:
        assert torch.equal(new[0], new[2])                          # the pixel stride of dY changes nothing
    again = run()
    assert torch.equal(new[0], again[0]) and (mode != "fwd" or torch.equal(new[1], again[1]))


@pytest.mark.parametrize("case", [(2, 40, 52), (1, 33, 32), (3, 64, 100), (2, 17, 257)])
def test_stem_bwd_weight_bn_fused(dev, case):
    """csrc/stem.hip stem_wgrad_bn_kernel (the stem's weight gradient with BatchNorm + LeakyReLU backward formed on the fly, fp32
    MFMA over padded positions) against the two calls it replaces (dcn_bn_act_bwd_apply + dcn_conv2d_bwd_weight) and against fp64
    autograd of conv -> batch norm (batch statistics) -> LeakyReLU."""
    from dcnet_amd import ops
    n, h, w = case
    img = _rand(n, 3, h, w, seed=91)
    wgt = (_rand(32, 3, 3, 3, seed=92) / 3)
    gamma = (torch.rand(32, generator=torch.Generator().manual_seed(93)) + 0.5); beta = _rand(32, seed=94) / 4
    dout_c = _rand(n, 32, h, w, seed=95)
    # fp64 truth
    wd_ = wgt.double().requires_grad_(True); g_ = gamma.double().requires_grad_(True); b_ = beta.double().requires_grad_(True)
    yd = F.conv2d(img.double(), wd_, padding=1)
    out = F.leaky_relu(F.batch_norm(yd, None, None, g_, b_, True, 0.1, 1e-5), 0.1)
    out.backward(dout_c.double())
    # device: forward pieces, then the two backward paths
    x = ops.nchw_to_nhwc(img.to(dev), 4)
    w_ohwi = ops.weight_to_ohwi(wgt.to(dev))
    y, stats = ops.conv2d_fwd(x, w_ohwi, 3, 1, want_stats=True)
    rm = torch.zeros(32, device=dev); rv = torch.ones(32, device=dev)
    mi = ops.bn_finalize(stats, n * h * w, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rm, rv)
    dout = _nhwc(dout_c).to(dev)
    dy, dgamma, dbeta = ops.bn_act_bwd(y, dout, mi[0], mi[1], gamma.to(dev), beta.to(dev), ops.ACT_LEAKY, 0.1)
    dw_old = ops.conv2d_bwd_weight(x, dy, 3, 1)
    dw_new, dgamma2, dbeta2 = ops.stem_bwd_weight_bn(x, y, dout, mi[0], mi[1], gamma.to(dev), beta.to(dev), ops.ACT_LEAKY, 0.1)
    assert dw_new.shape == dw_old.shape == (32, 64)
    assert torch.equal(dgamma, dgamma2) and torch.equal(dbeta, dbeta2)
    _close(dw_new, dw_old, 2e-5, "fused vs apply + wgrad")
    assert float(dw_new[:, 36:].abs().max()) == 0 and float(dw_new[:, :36].view(32, 9, 4)[..., 3].abs().max()) == 0
    got = ops.weight_grad_to_oihw(dw_new, (32, 3, 3, 3))
    _close(got, wd_.grad, 1e-4, "fused stem weight gradient vs fp64 autograd")
    _close(dgamma2, g_.grad, 1e-4, "dgamma"); _close(dbeta2, b_.grad, 1e-4, "dbeta")
    assert torch.equal(dw_new, ops.stem_bwd_weight_bn(x, y, dout, mi[0], mi[1], gamma.to(dev), beta.to(dev), ops.ACT_LEAKY, 0.1)[0])


def test_nconv_kernels_on_tensors_beyond_2_gib(dev):
    """The register-bank kernels address their operands relative to each workgroup's first row, so tensors of more than 2^31 bytes
    (configs[3] / configs[4] geometries: 64 x 608 x 608 x 32 floats = 3 GB) stay on them: forward 32 -> 64 stride 2 and its data
    gradient (with the BatchNorm tap) on 48 images of 608 x 608 against the implicit-GEMM tiles."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w = 48, 608, 608
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(n, h, w, 32, device=dev)
    assert x.numel() * 4 > 2 ** 31
    wgt = (torch.randn(64, 32, 3, 3, generator=g) / 6)
    w_ohwi = wgt.permute(0, 2, 3, 1).contiguous().to(dev)
    dy = torch.randn(n, h // 2, w // 2, 64, device=dev)
    try:
        lib().set_tuning(b"Nconv", 0)
        y0, st0 = ops.conv2d_fwd(x, w_ohwi, 3, 2, want_stats=True)
        dx0 = ops.conv2d_bwd_data(dy, w_ohwi, (h, w), 3, 2)
        lib().set_tuning(b"Nconv", 1)
        lib().prof_enable(1)
        y1, st1 = ops.conv2d_fwd(x, w_ohwi, 3, 2, want_stats=True)
        dx1 = ops.conv2d_bwd_data(dy, w_ohwi, (h, w), 3, 2)
        lib().prof_enable(0)
        ran = _prof_launches(37) + _prof_launches(38)
    finally:
        lib().set_tuning(b"Nconv", 1); lib().prof_enable(0)
    assert ran == 2, "the register-bank kernels declined the large tensors"
    assert float((y1 - y0).abs().max()) <= 2e-5 * float(y0.abs().max())
    assert float((dx1 - dx0).abs().max()) <= 2e-5 * float(dx0.abs().max())
    s0, s1 = st0.double().sum(0), st1.double().sum(0)
    assert float((s1 - s0).abs().max()) <= 1e-4 * float(s0.abs().max())
    for t in (y1[-1], y1[0], dx1[-1], dx1[0]):                      # both ends of the tensors were written
        assert float(t.abs().max()) > 0


@pytest.mark.parametrize("case", [(3, 20, 169, 512), (2, 7, 676, 256), (4, 20, 2704, 512), (1, 32, 100, 96)])
def test_k14_dlag_grouped_positions(dev, case):
    """csrc/sample.hip k14_dlag_kernel (gradient of the language side of Crossmodal_corrspondence, model/DCNet_model.py:41-112: the
    per-word sums of d_k over the positions whose arg-max word it is, then the backward of the column normalisation) against the
    same formula in torch fp64, with words that own no position and words that own all of them."""
    from dcnet_amd import ops
    n, L, hw, e = case
    g = torch.Generator().manual_seed(17)
    lag = torch.randn(n, L, e, generator=g); ln = torch.rand(n, e, generator=g) + 0.1
    cols = torch.randint(0, L, (n, hw), generator=g)
    cols[0] = 3 % L                                                 # one image: every position on the same word
    d_k = torch.randn(n, hw, e, generator=g)
    got = ops.k14_dlag(lag.to(dev), ln.to(dev), cols.to(dev), d_k.to(dev))
    dlag = torch.zeros(n, L, e, dtype=torch.float64)
    for i in range(n):
        dlag[i].index_add_(0, cols[i], d_k[i].double())
    dot = (dlag * lag.double()).sum(1, keepdim=True)
    want = (dlag - lag.double() * dot) / ln.double().clamp_min(1e-12).unsqueeze(1)
    assert got.shape == (n, L, 2 * e)
    _close(got[..., 0::2], want, 2e-5, "k14_dlag")
    assert float(got[..., 1::2].abs().max()) == 0
    assert torch.equal(got, ops.k14_dlag(lag.to(dev), ln.to(dev), cols.to(dev), d_k.to(dev)))


@pytest.mark.parametrize("case", [(3, 169, 64, 30, 10), (2, 676, 128, 30, 10), (2, 50, 32, 7, 3)])
def test_k9_bwd_match_lists(dev, case):
    """csrc/sample.hip k9_bwd_kernel (gradient of the K9 gathers of Interframe_corrspondence, model/DCNet_model.py:381-430, by
    destination position) against index_add in fp64: repeated positions, positions hit by both lists, untouched positions."""
    from dcnet_amd import ops
    b, hw, e, top_k, neg_n = case
    g = torch.Generator().manual_seed(23)
    index = torch.randint(0, hw * hw, (b, top_k), generator=g)
    index[0, :4] = index[0, 0]                                      # the same (i, j) picked several times
    neg_idx = torch.randint(0, hw, (b, top_k, neg_n), generator=g)
    neg_idx[0, 0, :] = int(index[0, 0]) % hw                        # negatives on a position of the direct list
    d_frame = torch.randn(b, top_k, e, generator=g); d_corr = torch.randn(b, top_k, e, generator=g)
    d_neg = torch.randn(b, top_k, neg_n, e, generator=g)
    got = ops.k9_bwd(index.to(dev), neg_idx.to(dev), d_frame.to(dev), d_corr.to(dev), d_neg.to(dev), hw)
    want = torch.zeros(2 * b, hw, e, dtype=torch.float64)
    for p in range(b):
        want[2 * p].index_add_(0, index[p] // hw, d_frame[p].double())
        want[2 * p + 1].index_add_(0, index[p] % hw, d_corr[p].double())
        want[2 * p + 1].index_add_(0, neg_idx[p].reshape(-1), d_neg[p].reshape(-1, e).double())
    _close(got, want, 1e-5, "k9_bwd")
    assert torch.equal(got, ops.k9_bwd(index.to(dev), neg_idx.to(dev), d_frame.to(dev), d_corr.to(dev), d_neg.to(dev), hw))


def _prof_launches(tag):
    """launches booked under a profiling tag since dcn_prof_enable(1) (csrc/prof.h)"""
    import ctypes
    from dcnet_amd.lib import lib
    c = (ctypes.c_int64 * 64)(); m = (ctypes.c_double * 64)(); wk = (ctypes.c_double * 64)()
    lib().prof_collect(ctypes.addressof(c), ctypes.addressof(m), ctypes.addressof(wk), 0)
    return c[tag]


W9_CASES = [
    # n, h, w, stride, cin   (3x3, cin input channels, 2 cin filters; output rows of >= 32 pixels)
    (2, 64, 64, 1, 32),
    (2, 40, 70, 1, 32),      # rows that are no multiple of the 32-position K-step
    (1, 35, 131, 1, 32),     # odd map
    (4, 33, 33, 1, 32),      # shortest rows (34 padded entries: a wrap in every K-step), image-to-image wrap
    (2, 64, 64, 2, 32),      # 32-pixel output rows
    (1, 72, 136, 2, 32),
    (3, 66, 96, 2, 32),
    (2, 64, 64, 1, 64),      # 64 -> 128: eight waves, a wave = 32 filters x 32 channels x all nine taps
    (1, 35, 131, 1, 64),
    (4, 33, 33, 1, 64),
    (2, 64, 64, 2, 64),
    (3, 66, 96, 2, 64),
]


@pytest.mark.parametrize("case", W9_CASES)
def test_wgrad9_nine_tap_kernel(dev, case):
    """csrc/wgrad9.hip (weight gradient of the 32 -> 64 and 64 -> 128 3x3 layers of the 416 / 208 / 104 maps, stride 1 and 2, all nine
    taps per workgroup, K over padded positions) against fp64 and against the kernels it replaces: image borders (pads must read as
    zero, top / bottom filter rows masked), the stride-2 even / odd planes, split-K slabs and a dY that is a slice of a wider tensor."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, st, cin = case
    cout = 2 * cin
    x = _rand(n, h, w, cin, seed=51).to(dev)
    wide = (_rand(n, h // st, w // st, cout + 32, seed=52) / 8).to(dev)
    dy = wide[..., 16:16 + cout]                                   # pixel stride cout + 32
    xd = x.permute(0, 3, 1, 2).double().cpu()
    wgt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, wgt, padding=1, stride=st).backward(dy.permute(0, 3, 1, 2).double().cpu())
    ref = wgt.grad.permute(0, 2, 3, 1)                             # OHWI
    try:
        lib().set_tuning(b"9tap", 0)
        old = ops.conv2d_bwd_weight(x, dy, 3, st)
        lib().set_tuning(b"9tap", 3)                               # (3: the 64 -> 128 form at both strides)
        lib().prof_enable(1)
        new = ops.conv2d_bwd_weight(x, dy, 3, st)
        lib().prof_enable(0)
        ran = _prof_launches(36)
        new_c = ops.conv2d_bwd_weight(x, dy.contiguous(), 3, st)
        for target in (1, 7, 4096):                                # one slab ... as many as the positions allow
            lib().set_tuning(b"9target", target)
            alt = ops.conv2d_bwd_weight(x, dy, 3, st)
            _close(alt, ref, 3e-5, f"wgrad9 target {target}")
        lib().set_tuning(b"9target", 512)
        again = ops.conv2d_bwd_weight(x, dy, 3, st)
    finally:
        lib().set_tuning(b"9tap", W9_DEFAULT); lib().set_tuning(b"9target", 512); lib().prof_enable(0)
    assert ran == 1, "the nine-tap kernel did not run"
    _close(new, ref, 3e-5, "wgrad9")
    _close(old, ref, 3e-5, "the kernel it replaces")
    assert torch.equal(new, new_c)                                 # the pixel stride of dY changes nothing
    assert torch.equal(new, again)                                 # fixed summation order


W9_DEFAULT = 2      # csrc/wgrad9.hip g_w9


W1X_CASES = [
    # n, h, w, cin, cout   (1x1 stride 1, >= 256 input channels, > 128 filters, >= 4096 pixels)
    (8, 26, 26, 512, 256),
    (2, 52, 52, 256, 160),     # one ragged filter tile
    (8, 26, 26, 256, 512),
    (7, 25, 27, 384, 192),     # ragged tiles on both sides, odd pixel count
    (32, 13, 13, 1024, 512),
]


@pytest.mark.parametrize("case", W1X_CASES)
def test_wgrad1x_wide_tile_kernel(dev, case):
    """csrc/wgrad.hip wgrad1x_kernel (weight gradient of the 1x1 stride-1 layers on a 256-wide tile, eight waves, rotating fragment set)
    against fp64 and against the 128 x 128 tile it replaces, with a dY that is a slice of a wider tensor, several split-K targets, and
    twice in a row (fixed summation order)."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout = case
    x = _rand(n, h, w, cin, seed=51).to(dev)
    wide = (_rand(n, h, w, cout + 32, seed=52) / 8).to(dev)
    dy = wide[..., 16:16 + cout]
    ref = torch.einsum("nhwo,nhwi->oi", dy.double().cpu(), x.double().cpu()).view(cout, 1, 1, cin)
    try:
        lib().set_tuning(b"Y1wide", 0)
        old = ops.conv2d_bwd_weight(x, dy, 1, 1)
        lib().set_tuning(b"Y1wide", 1)
        new = ops.conv2d_bwd_weight(x, dy, 1, 1)
        again = ops.conv2d_bwd_weight(x, dy.contiguous(), 1, 1)
        for target in (64, 1024):
            lib().set_tuning(b"Y1wide", target)
            alt = ops.conv2d_bwd_weight(x, dy, 1, 1)
            _close(alt, ref, 3e-5, f"wgrad1x target {target}")
    finally:
        lib().set_tuning(b"Y1wide", 1)
    _close(new, ref, 3e-5, "wgrad1x")
    _close(old, ref, 3e-5, "128 x 128 tile")
    _close(new, old, 3e-6, "wgrad1x vs the 128 x 128 tile")
    assert torch.equal(new, again)
    assert not torch.equal(new, old), "wgrad1x_kernel did not run"


W3_CASES = [
    # n, h, w, cin, cout   (3x3 stride 1, >= 128 channels both sides, >= 1024 pixels)
    (8, 13, 13, 128, 256),
    (4, 26, 26, 256, 128),
    (2, 33, 31, 136, 160),     # ragged channel tiles, odd map
    (3, 20, 45, 128, 128),     # wide rows
    (2, 52, 52, 128, 128),
    (9, 11, 12, 192, 320),     # short rows: two padded rows per 16-position K-step
    (2, 40, 40, 64, 128),      # 64 input channels: half of the channel tile empty
    (2, 40, 40, 128, 64),      # 64 filters: half of the filter tile empty
]


@pytest.mark.parametrize("m16", [1, 0])
@pytest.mark.parametrize("case", W3_CASES)
def test_wgrad3_filter_row_kernel(dev, case, m16):
    """csrc/wgrad3.hip (weight gradient of the 3x3 stride-1 layers, one filter row per workgroup, K over padded pixel
    coordinates) against fp64 and against the per-tap kernel it replaces, including a dY that is a slice of a wider tensor.
    Image borders (no masks: the pads must read as zero), image-to-image wrap of the strip, split-K slabs and ragged channel
    tiles all matter here."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout = case
    x = _rand(n, h, w, cin, seed=41).to(dev)
    wide = (_rand(n, h, w, cout + 32, seed=42) / 8).to(dev)
    dy = wide[..., 16:16 + cout]                                   # pixel stride cout + 32
    xd = x.permute(0, 3, 1, 2).double().cpu()
    wgt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, wgt, padding=1).backward(dy.permute(0, 3, 1, 2).double().cpu())
    ref = wgt.grad.permute(0, 2, 3, 1)                             # OHWI
    try:
        lib().set_tuning(b"U3m16", m16)                            # 16x16x32 MFMAs, 32 positions per K-step (1) / 32x32x16, 16 positions (0, the default)
        lib().set_tuning(b"u3row", 0)
        old = ops.conv2d_bwd_weight(x, dy, 3, 1)
        lib().set_tuning(b"u3row", 1)
        new = ops.conv2d_bwd_weight(x, dy, 3, 1)
        new_c = ops.conv2d_bwd_weight(x, dy.contiguous(), 3, 1)
        if m16:
            lib().set_tuning(b"U3m16", 0)
            other = ops.conv2d_bwd_weight(x, dy, 3, 1)
            lib().set_tuning(b"U3m16", 1)
            assert not torch.equal(new, other), "the 16x16x32 build did not run"
            _close(new, other, 3e-6, "wgrad3: 16x16x32 vs 32x32x16")
        for target in (96, 2048):                                  # one split ... many splits
            lib().set_tuning(b"v3target", target)
            alt = ops.conv2d_bwd_weight(x, dy, 3, 1)
            _close(alt, ref, 3e-5, f"wgrad3 target {target}")
    finally:
        lib().set_tuning(b"u3row", 1); lib().set_tuning(b"v3target", 512); lib().set_tuning(b"U3m16", 0)
    # bf16-operand mode (configs[2]; also the weight gradient of the fp8 mode): the same kernel with one bf16 plane per operand,
    # against its exact model — the fp64 weight gradient of the bf16-rounded tensors
    rb = lambda t: t.to(torch.bfloat16).double()
    wgt16 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(rb(x.cpu()).permute(0, 3, 1, 2), wgt16, padding=1).backward(rb(dy.cpu()).permute(0, 3, 1, 2))
    ref16 = wgt16.grad.permute(0, 2, 3, 1)
    try:
        ops.set_precision("bf16")
        got16 = ops.conv2d_bwd_weight(x, dy, 3, 1)
        lib().set_tuning(b"u3row", 0)
        old16 = ops.conv2d_bwd_weight(x, dy, 3, 1)
    finally:
        lib().set_tuning(b"u3row", 1)
        ops.set_precision("fp32")
    _close(got16, ref16, 3e-5, "wgrad3 bf16 operands vs exact model")
    _close(old16, ref16, 3e-5, "per-tap bf16 operands vs exact model")     # (64-channel sides run on the 128x128 tile as well)
    assert float((got16.double().cpu() - ref).abs().max()) > 1e-5 * max(1.0, float(ref.abs().max()))      # it IS reduced precision
    _close(new, ref, 3e-5, "wgrad3")
    _close(old, ref, 3e-5, "per-tap wgrad")
    _close(new, old, 3e-6, "wgrad3 vs per-tap")
    assert torch.equal(new, new_c)                                 # the pixel stride of dY changes nothing
    assert not torch.equal(new, old), "the filter-row kernel did not run"


SLAB_FOLD_CASES = [
    # n, h, w, cin, cout, k, stride, knobs  — every kernel family that folds (csrc/slabsum.h), ragged tiles, many and few splits
    (8, 13, 13, 128, 256, 3, 1, {}),                      # wgrad3_kernel<2>
    (2, 33, 31, 136, 160, 3, 1, {}),                      # ... ragged channel tiles
    (9, 11, 12, 192, 320, 3, 1, {b"U3m16": 1}),           # wgrad3x_kernel
    (8, 26, 26, 256, 512, 1, 1, {}),                      # wgrad1x_kernel<4>
    (7, 25, 27, 384, 192, 1, 1, {}),                      # ... ragged
    (4, 26, 26, 256, 128, 1, 1, {}),                      # wgrad_kernel<128,128> f16 split
    (8, 54, 58, 64, 128, 3, 2, {}),                       # ... stride 2, nine taps
    (2, 16, 20, 64, 32, 1, 1, {}),                        # narrow fp32-pipe tile, WK > 1 (waves that own no output rows)
    (1, 32, 32, 4, 32, 3, 1, {}),                         # stem layout (c4)
]


@pytest.mark.parametrize("case", SLAB_FOLD_CASES)
def test_slab_fold_is_bitwise_the_separate_pass(dev, case):
    """csrc/slabsum.h: the split-K slabs summed by the last-arriving workgroup of each tile must be the sum reduce_slabs_kernel
    forms behind the launch, bit for bit, whichever workgroup arrives last; the arrival counters must be zero again afterwards
    (the next launch and every replay of a captured step start from them); fp32 and bf16 storage."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    n, h, w, cin, cout, k, s, knobs = case
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    x = _rand(n, h, w, cin, seed=61).to(dev)
    dy = (_rand(n, ho, wo, cout, seed=62) / 8).to(dev)
    try:
        for kk, vv in knobs.items():
            lib().set_tuning(kk, vv)
        lib().set_tuning(b"Slabfold", 0)
        sep = ops.conv2d_bwd_weight(x, dy, k, s)
        lib().set_tuning(b"Slabfold", 1 << 20)               # no size limit: every launch with more than one split folds
        folds = [ops.conv2d_bwd_weight(x, dy, k, s) for _ in range(3)]
        if cin % 8 == 0 and cout % 8 == 0:
            xb, dyb = x.to(torch.bfloat16), dy.to(torch.bfloat16)
            fold16 = ops.conv2d_bwd_weight_b16(xb, dyb, k, s)
            lib().set_tuning(b"Slabfold", 0)
            sep16 = ops.conv2d_bwd_weight_b16(xb, dyb, k, s)
            assert torch.equal(fold16, sep16)
    finally:
        lib().set_tuning(b"Slabfold", 0)                     # (the default: off — measured slower in the step, csrc/slabsum.h)
        for kk in knobs:
            lib().set_tuning(kk, 0)
    for f in folds:
        assert torch.equal(f, sep)
    assert int(ops.slab_counters(dev, 0).abs().sum()) == 0
    nws = lib().conv2d_bwd_weight_ws(n, h, w, cin, cout, k, s)
    assert nws > 0, "the case has one split: nothing was folded"
