"""fp8 storage (BASELINE.json configs[4]; ``ops.set_precision("fp8s")``): the forward and data-gradient operands of the convolutions ARE
1-byte OCP e4m3 tensors in HBM, one e8m0 scale per row (pixel / filter), multiplied by the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4.
The reference has no fp8 semantics, so parity is defined against the EXACT MODEL: the quantiser against torch's
float8_e4m3fn on the scaled rows (bit for bit), the convolutions against fp64 on the dequantised operands, rounded where the kernel
stores bf16.  The block-scaled MFMA does not accumulate like an fp32 FMA chain: measured (tools/f8_diag.py) its results sit up to 3e-5 of
SUM |a_k b_k| away from fp64 (median 7e-7) whatever the scales — its adder tree keeps about 15 bits below the largest product — so the
comparisons allow 1e-4 of that sum on top of the one bf16 rounding of a stored result."""
import pytest
import torch
import torch.nn.functional as F

from test_b16_gpu import _bf, _rand, _ulp_close

pytestmark = pytest.mark.gpu


def _close(got, ref64, sumabs64, name, stored_bf16=True):
    """got against the fp64 exact model: one bf16 rounding (if stored as bf16) plus 1e-4 of the sum of product magnitudes of each output"""
    g = got.double().cpu(); r = ref64.double().cpu()
    assert g.shape == r.shape, (name, g.shape, r.shape)
    tol = (1.01 * 2.0 ** -8 if stored_bf16 else 2.0 ** -22) * r.abs() + 1e-4 * sumabs64.double().cpu() + 1e-30
    bad = (g - r).abs() > tol
    assert not bool(bad.any()), f"{name}: {int(bad.sum())} of {bad.numel()} beyond the model; worst excess {float(((g - r).abs() - tol).max()):.3e}"


def _mx_rows(x16):
    """exact model of dcn_quant_rows_e4m3: (bytes, scale bytes, dequantised fp64) of bf16 rows x16 (..., c)"""
    xf = x16.float().cpu()
    amax = xf.abs().amax(-1, keepdim=True)
    mant, ex = torch.frexp(amax)                         # amax = mant * 2^ex, mant in [0.5, 1): floor(log2(amax)) = ex - 1
    e = torch.where(amax > 0, ex - 1 - 8, torch.zeros_like(ex)).clamp(-126, 126)
    scaled = (xf * torch.exp2(-e.float())).clamp(-448.0, 448.0)
    q = scaled.to(torch.float8_e4m3fn)
    return q.view(torch.uint8), (e + 127).to(torch.uint8).reshape(-1), q.double() * torch.exp2(e.double())


@pytest.mark.parametrize("shape", [(37, 64), (5, 7, 128), (1000, 256), (3, 1024), (129, 2304), (64, 8)])
def test_quantiser_matches_torch_float8_bit_for_bit(shape):
    from dcnet_amd import ops
    dev = torch.device("cuda:0")
    x = _bf(_rand(*shape, seed=3) * torch.logspace(-3, 2, shape[0]).reshape(-1, *([1] * (len(shape) - 1)))).contiguous()
    x.view(-1, shape[-1])[1].zero_()                     # an all-zero row: scale byte 127, zeros
    x.view(-1, shape[-1])[2, 0] = 3.0e4                   # a row whose maximum dwarfs the rest (small entries land in e4m3's subnormals / zero)
    q, s = ops.quant_rows_e4m3(x.to(dev))
    rq, rs, _ = _mx_rows(x)
    assert torch.equal(s.cpu(), rs), (s.cpu()[:8], rs[:8])
    assert torch.equal(q.cpu(), rq.reshape(q.shape))
    assert int(s[1]) == 127 and int(q.view(-1, shape[-1])[1].max()) == 0


CASES = [
    # n, h, w, cin, cout, k, stride
    (2, 13, 13, 64, 128, 3, 1),
    (2, 13, 13, 128, 64, 1, 1),
    (1, 9, 11, 192, 160, 3, 1),        # ragged M, 160 = 5 x 32 filters, three K-steps per tap
    (3, 8, 8, 256, 256, 1, 1),
    (1, 27, 29, 64, 128, 3, 2),        # odd sizes under stride 2 (ragged parity classes)
    (2, 52, 52, 128, 256, 3, 1),       # more than one round of tiles (the 128 x 256 tile)
    (2, 26, 26, 256, 128, 3, 1),
    (8, 13, 13, 256, 512, 3, 1),
    (1, 28, 28, 128, 256, 3, 2),
    (1, 20, 20, 256, 64, 3, 2),        # 64 filters: the 128 x 64 tile
]


@pytest.mark.parametrize("case", CASES)
def test_f8_conv_forward_and_data_gradient_match_their_exact_model(case):
    from dcnet_amd import ops
    dev = torch.device("cuda:0")
    n, h, w, cin, cout, k, st = case
    T = k * k
    # per-pixel magnitudes over three decades, per-filter magnitudes over two: the row scales must do their work
    mag = torch.logspace(-2, 1, n * h * w)[torch.randperm(n * h * w, generator=torch.Generator().manual_seed(5))].reshape(n, h, w, 1)
    x = _bf(_rand(n, h, w, cin, seed=1) * mag)
    wt = _bf(_rand(cout, k, k, cin, seed=2) / (cin * T) ** 0.5 * torch.logspace(-1, 1, cout).reshape(cout, 1, 1, 1))
    x8, xs = ops.quant_rows_e4m3(x.to(dev))
    w8, ws = ops.quant_rows_e4m3(wt.reshape(cout, T * cin).to(dev))
    _, _, xd = _mx_rows(x)
    _, _, wd = _mx_rows(wt.reshape(cout, T * cin))
    xd = xd.permute(0, 3, 1, 2).requires_grad_(True)
    wdq = wd.reshape(cout, k, k, cin).permute(0, 3, 1, 2)
    yd = F.conv2d(xd, wdq, stride=st, padding=(k - 1) // 2)
    ref_y = yd.detach().permute(0, 2, 3, 1)
    abs_y = F.conv2d(xd.detach().abs(), wdq.abs(), stride=st, padding=(k - 1) // 2).permute(0, 2, 3, 1)      # sum of |products| per output
    ho, wo = yd.shape[2], yd.shape[3]
    # ---- forward: raw result bf16 + BatchNorm partial sums of the stored values; fp32 output; epilogue ----
    y, stats = ops.conv2d_fwd_f8(x8, xs, w8.reshape(-1), ws, cout, k, st, want_stats=True)
    assert y.dtype == torch.bfloat16 and y.shape == (n, ho, wo, cout)
    _close(y, ref_y, abs_y, "fwd")
    s = stats.double().sum(0).cpu()
    yf = y.double().cpu().reshape(-1, cout)
    assert torch.allclose(s[0], yf.sum(0), rtol=1e-5, atol=1e-4 * float(yf.abs().sum(0).max()))
    assert torch.allclose(s[1], (yf * yf).sum(0), rtol=1e-5, atol=1e-6 * float((yf * yf).sum(0).max()))
    y32, _ = ops.conv2d_fwd_f8(x8, xs, w8.reshape(-1), ws, cout, k, st, out_f32=True)
    _close(y32, ref_y, abs_y, "fwd fp32 out", stored_bf16=False)
    assert torch.equal(ops.conv2d_fwd_f8(x8, xs, w8.reshape(-1), ws, cout, k, st)[0], y)                 # bitwise repeatable
    sc = (_rand(cout, seed=4).abs() + 0.5).to(dev); sh = _rand(cout, seed=5).to(dev)
    res = _bf(_rand(n, ho, wo, cout, seed=6)).to(dev)
    o, _ = ops.conv2d_fwd_f8(x8, xs, w8.reshape(-1), ws, cout, k, st, sc, sh, ops.ACT_LEAKY, 0.1, residual=res)
    t = ref_y * sc.double().cpu() + sh.double().cpu()
    _close(o, torch.where(t > 0, t, 0.1 * t) + res.double().cpu(), abs_y * sc.double().cpu(), "fwd epilogue")
    # ---- data gradient: dy quantised per pixel, the transposed bank [Cin][T][Cout] per row ----
    if cout % 64:
        return
    dy = _bf(_rand(n, ho, wo, cout, seed=3) / 8 * torch.logspace(-2, 0, n * ho * wo).reshape(n, ho, wo, 1))
    wt_t = wt.reshape(cout, T, cin).permute(2, 1, 0).contiguous().reshape(cin, T * cout)
    dy8, dys = ops.quant_rows_e4m3(dy.to(dev))
    wt8, wts = ops.quant_rows_e4m3(wt_t.to(dev))
    _, _, dyd = _mx_rows(dy)
    _, _, wtd = _mx_rows(wt_t)
    w_for_dgrad = wtd.reshape(cin, k, k, cout).permute(3, 0, 1, 2)                    # OIHW from the dequantised transposed bank
    xg = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
    F.conv2d(xg, w_for_dgrad, stride=st, padding=(k - 1) // 2).backward(dyd.permute(0, 3, 1, 2))
    ref_dx = xg.grad.permute(0, 2, 3, 1)
    xg2 = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
    F.conv2d(xg2, w_for_dgrad.abs(), stride=st, padding=(k - 1) // 2).backward(dyd.abs().permute(0, 3, 1, 2))
    abs_dx = xg2.grad.permute(0, 2, 3, 1)
    dx = ops.conv2d_bwd_data_f8(dy8, dys, wt8.reshape(-1), wts, (h, w), cin, k, st)
    _close(dx, ref_dx, abs_dx, "dgrad")
    base = _bf(_rand(n, h, w, cin, seed=7)).to(dev)
    acc = base.clone()
    ops.conv2d_bwd_data_f8(dy8, dys, wt8.reshape(-1), wts, (h, w), cin, k, st, out=acc, accumulate=True)
    _close(acc, ref_dx + base.double().cpu(), abs_dx, "dgrad accumulate")
    assert torch.equal(ops.conv2d_bwd_data_f8(dy8, dys, wt8.reshape(-1), wts, (h, w), cin, k, st), dx)
    dx32 = ops.conv2d_bwd_data_f8(dy8, dys, wt8.reshape(-1), wts, (h, w), cin, k, st, out_f32=True)
    _close(dx32, ref_dx, abs_dx, "dgrad fp32 out", stored_bf16=False)
    if st == 1:
        # the BatchNorm tap of the stride-1 data gradient against the bf16 reduce pass on the finished dx
        yprev = _bf(_rand(n, h, w, cin, seed=8)).to(dev)
        mean = _rand(cin, seed=9).to(dev) * 0.1; invstd = (_rand(cin, seed=10).abs() + 0.5).to(dev)
        gamma = (_rand(cin, seed=11).abs() + 0.5).to(dev); beta = _rand(cin, seed=12).to(dev)
        tap = dict(y=yprev, mean=mean, invstd=invstd, gamma=gamma, beta=beta, act=ops.ACT_LEAKY, slope=0.1)
        dxt, part = ops.conv2d_bwd_data_f8(dy8, dys, wt8.reshape(-1), wts, (h, w), cin, k, st, tap=tap)
        assert torch.equal(dxt, dx) and part is not None
        xh = (yprev.double().cpu() - mean.double().cpu()) * invstd.double().cpu()
        g = dx.double().cpu() * torch.where(gamma.double().cpu() * xh + beta.double().cpu() <= 0, 0.1, 1.0)
        sums = part.double().sum(0).cpu()
        assert torch.allclose(sums[0], g.reshape(-1, cin).sum(0), rtol=1e-4, atol=1e-4 * float(g.abs().reshape(-1, cin).sum(0).max()))
        assert torch.allclose(sums[1], (g * xh).reshape(-1, cin).sum(0), rtol=1e-4, atol=1e-4 * float((g * xh).abs().reshape(-1, cin).sum(0).max()))


@pytest.mark.parametrize("c", [64, 128, 256, 512])
def test_fused_quantisers_equal_the_pass_on_the_stored_tensor(c):
    """scale_act / bn_act_bwd with quant=True write the e4m3 copy of their bf16 result in the same pass: bit for bit what
    dcn_quant_rows_e4m3 makes of that result, and the bf16 result itself unchanged."""
    from dcnet_amd import ops
    dev = torch.device("cuda:0")
    n, h, w = 3, 9, 11
    y = _bf(_rand(n, h, w, c, seed=1) * torch.logspace(-2, 1, n * h * w).reshape(n, h, w, 1)).to(dev)
    res = _bf(_rand(n, h, w, c, seed=2)).to(dev)
    sc = (_rand(c, seed=3).abs() + 0.5).to(dev); sh = _rand(c, seed=4).to(dev)
    for r_ in (None, res):
        plain = ops.scale_act(y, sc, sh, ops.ACT_LEAKY, 0.1, residual=r_)
        fused = ops.scale_act(y, sc, sh, ops.ACT_LEAKY, 0.1, residual=r_, quant=True)
        assert torch.equal(plain, fused) and getattr(fused, "_dcn_q8", None) is not None
        q, s = ops.quant_rows_e4m3(plain)
        assert torch.equal(fused._dcn_q8[0], q) and torch.equal(fused._dcn_q8[1], s)
        assert ops.quant_of(fused)[0] is fused._dcn_q8[0]
    dout = _bf(_rand(n, h, w, c, seed=5) / 16).to(dev)
    mean = (_rand(c, seed=6) * 0.1).to(dev); invstd = (_rand(c, seed=7).abs() + 0.5).to(dev)
    dy0, dg0, db0 = ops.bn_act_bwd(y, dout, mean, invstd, sc, sh, ops.ACT_LEAKY, 0.1)
    dy1, dg1, db1 = ops.bn_act_bwd(y, dout, mean, invstd, sc, sh, ops.ACT_LEAKY, 0.1, quant=True)
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    nd = int((dy0 != dy1).sum())
    assert nd == 0, (nd, float((dy0.float() - dy1.float()).abs().max()), float(dy0.float().abs().max()))
    q, s = ops.quant_rows_e4m3(dy0)
    assert torch.equal(dy1._dcn_q8[0], q) and torch.equal(dy1._dcn_q8[1], s)


def test_banks_quantised_once_per_refresh_equal_the_per_layer_pass():
    """fp8 storage: FilterBanks.refresh makes the e4m3 forms of every 3x3 bank the mode reads (forward and transposed) in ONE launch
    (csrc/b16.hip quant_banks_kernel) — bit for bit what a quant_rows_e4m3 pass per layer and direction made in front of each convolution;
    layers the mode does not take (1x1, fewer than F8_MIN_K channels on the contraction side) get none; a second refresh after the weights
    moved follows them."""
    from dcnet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    ws = {0: torch.randn(256, 128, 3, 3, generator=g), 1: torch.randn(128, 256, 1, 1, generator=g), 2: torch.randn(96, 192, 3, 3, generator=g),
          3: torch.randn(64, 32, 3, 3, generator=g), 4: torch.randn(1024, 512, 3, 3, generator=g) * 1e-3}
    ws = {i: w.to(dev).contiguous() for i, w in ws.items()}
    try:
        ops.set_precision("fp8s")
        fb = ops.FilterBanks(ws, dev)
        for round_ in range(2):
            fb.refresh()
            for i, w in ws.items():
                it = fb.get(i, w)
                co, ci, kh, kw = w.shape
                for key, b16, rows, takes in (("q8", it["b16"], co, ops.f8_takes(ci, co, kh)), ("tq8", it["tb16"], ci, ops.f8_takes(co, ci, kh))):
                    if not takes:
                        assert it.get(key) is None, (i, key)
                        continue
                    q, sc = it[key]
                    rq, rsc = ops.quant_rows_e4m3(b16.view(rows, -1))
                    assert torch.equal(q, rq) and torch.equal(sc, rsc), (i, key, round_)
                    assert ops.bank_q8(it, key, b16, rows)[0] is q
            for w in ws.values():
                w.mul_(1.7).add_(0.01)                       # the optimiser's step
        assert fb.get(0, ws[0]).get("q8") is not None and fb.get(1, ws[1]).get("q8") is None
    finally:
        ops.set_precision("fp32")
