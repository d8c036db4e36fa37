"""bf16 storage (BASELINE.json configs[2]; ``ops.set_precision("bf16s")``): activations, raw conv outputs and their gradients ARE bf16
tensors in HBM.  The reference has no bf16 semantics, so parity is defined against each kernel's EXACT MODEL — the same
arithmetic in fp64 on the bf16 values the kernel reads, rounded to bf16 where the kernel stores bf16 — which the kernels must meet to
accumulation-order accuracy; the end-to-end tests then bound the mode against the fp32 run of the same network."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _bf(t):
    return t.to(torch.bfloat16)


def _ulp_close(got, ref64, name, ulps=1.01, floor=1e-6):
    """got (bf16 tensor) against an fp64 reference: within `ulps` bf16 roundings of it (2^-8 relative each) plus a floor."""
    g = got.double().cpu(); r = ref64.double().cpu()
    assert g.shape == r.shape, (name, g.shape, r.shape)
    tol = ulps * 2.0 ** -8 * r.abs() + floor * max(1.0, float(r.abs().max()))
    bad = (g - r).abs() > tol
    assert not bool(bad.any()), f"{name}: {int(bad.sum())} of {bad.numel()} beyond {ulps} bf16 ulp; worst {float(((g - r).abs() - tol).max()):.3e}"


CASES = [
    # n, h, w, cin, cout, k, stride
    (2, 13, 13, 64, 128, 3, 1),
    (2, 13, 13, 128, 64, 1, 1),
    (1, 26, 26, 32, 64, 3, 2),
    (2, 16, 20, 64, 32, 1, 1),        # 32 filters: the 256 x 32 tile
    (1, 9, 11, 96, 160, 3, 1),        # ragged M, 160 = 5 x 32 filters
    (3, 8, 8, 256, 256, 1, 1),
    (1, 27, 29, 64, 128, 3, 2),       # odd sizes under stride 2 (ragged parity classes)
    (2, 52, 52, 128, 256, 3, 1),      # more than one round of tiles; weight gradient by filter rows (wgrad3.hip, bf16 inputs)
    (2, 26, 26, 256, 128, 3, 1),      # filter rows again, two channel tiles
    (1, 40, 33, 160, 192, 3, 1),      # ragged channel tiles (128 + 32 | 128 + 64), odd width: pad crossings inside a K-step
    (8, 13, 13, 256, 512, 3, 1),      # 13-wide rows: two row wraps per K-step
    (1, 28, 28, 128, 256, 3, 2),      # stride 2 into 256 filters (the 256 x 256 tile with a strided gather)
    (1, 20, 20, 256, 256, 3, 2),      # ... and its parity classes into 256 channels (strided output)
    (2, 48, 48, 32, 64, 3, 1),        # the narrow layers: weight gradient with all nine taps per workgroup (wgrad9.hip, bf16 inputs)
    (1, 64, 80, 32, 64, 3, 2),        # ... at stride 2 (even / odd entry planes)
    (1, 64, 66, 64, 128, 3, 1),       # ... and its 64 -> 128 form
    (4, 128, 128, 32, 64, 3, 2),      # the stride-2 data gradient of the narrow layers on the register-bank kernel (nconv.hip dgrad2, bf16 form)
    (2, 128, 256, 64, 128, 3, 2),     # ... its 64 <- 128 form
    (4, 128, 128, 32, 64, 3, 1),      # forward 32 -> 64 and data gradient 64 -> 32 on the register-bank kernel (nconv.hip nconv1, bf16 form)
    (4, 256, 256, 32, 64, 3, 2),      # ... the stride-2 forward (even / odd entry planes)
]


@pytest.mark.parametrize("tile2b", [0, 1, 2])
@pytest.mark.parametrize("case", CASES)
def test_b16_conv_forward_dgrad_wgrad_match_their_exact_model(case, tile2b):
    """tile2b = 1: every launch with a multiple of 256 filters is forced onto conv2b.hip's 256 x 256 eight-wave tile (knob `2btile`: it takes
    them from 256 tiles on by default, which only the full-size layers reach); 0: conv1.hip's conv1b tiles; 2: the 3x3 stride-1 launches on
    conv3.hip's strip kernel in its bf16-storage form (knob `3h16`, off by default: it measured slower than the gathered tiles)."""
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    if tile2b == 1 and case[3] % 256 and case[4] % 256:
        pytest.skip("no launch of this case has a multiple of 256 filters")
    if tile2b == 2 and not (case[5] == 3 and case[6] == 1 and max(case[3], case[4]) > 64):
        pytest.skip("no 3x3 stride-1 launch with more than 64 filters in this case")
    lib().set_tuning(b"2btile", -1 if tile2b == 1 else 0)
    lib().set_tuning(b"3h16", 1 if tile2b == 2 else 0)
    try:
        _conv_case(case)
    finally:
        lib().set_tuning(b"2btile", 256)
        lib().set_tuning(b"3h16", 0)


def _conv_case(case):
    from dcnet_amd import ops
    from dcnet_amd.lib import lib
    dev = torch.device("cuda:0")
    n, h, w, cin, cout, k, st = case
    T = k * k
    x = _bf(_rand(n, h, w, cin, seed=1)).to(dev)
    wt = _bf(_rand(cout, k, k, cin, seed=2) / (cin * T) ** 0.5).to(dev)              # OHWI bank, bf16
    xd = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    wd = wt.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    yd = F.conv2d(xd, wd, stride=st, padding=(k - 1) // 2)
    ho, wo = yd.shape[2], yd.shape[3]
    dy = _bf(_rand(n, ho, wo, cout, seed=3) / 8).to(dev)
    yd.backward(dy.double().cpu().permute(0, 3, 1, 2))
    ref_y = yd.detach().permute(0, 2, 3, 1)
    # ---- forward: raw result bf16 + BatchNorm partial sums of the stored values ----
    y, stats = ops.conv2d_fwd_b16(x, wt.reshape(-1), cout, k, st, want_stats=True)
    assert y.dtype == torch.bfloat16 and y.shape == (n, ho, wo, cout)
    _ulp_close(y, ref_y, "fwd")
    s = stats.double().sum(0).cpu()
    yf = y.double().cpu().reshape(-1, cout)
    assert torch.allclose(s[0], yf.sum(0), rtol=1e-5, atol=1e-4 * float(yf.abs().sum(0).max()))
    assert torch.allclose(s[1], (yf * yf).sum(0), rtol=1e-5, atol=1e-6 * float((yf * yf).sum(0).max()))
    y32, _ = ops.conv2d_fwd_b16(x, wt.reshape(-1), cout, k, st, out_f32=True)
    assert y32.dtype == torch.float32 and float((y32.double().cpu() - ref_y).abs().max()) <= 3e-5 * max(1.0, float(ref_y.abs().max()))
    again, _ = ops.conv2d_fwd_b16(x, wt.reshape(-1), cout, k, st)
    assert torch.equal(again, y)                                                      # bitwise repeatable
    lib().set_tuning(b"Nb16", 0)               # (the register-bank kernel of the 32 <-> 64 layers off: conv1.hip's gathered tiles)
    try:
        y0, stats0 = ops.conv2d_fwd_b16(x, wt.reshape(-1), cout, k, st, want_stats=True)
        _ulp_close(y0, ref_y, "fwd, gathered tiles")
        assert torch.allclose(stats0.double().sum(0).cpu()[0], yf.sum(0), rtol=1e-5, atol=1e-4 * float(yf.abs().sum(0).max())) or not torch.equal(y0, y)
        _ulp_close(ops.conv2d_bwd_data_b16(_bf(_rand(n, ho, wo, cout, seed=3) / 8).to(dev), wt.reshape(cout, T, cin).permute(2, 1, 0).contiguous().reshape(-1),
                                           (h, w), cin, k, st), xd.grad.permute(0, 2, 3, 1), "dgrad, gathered tiles")
    finally:
        lib().set_tuning(b"Nb16", 1)
    # ---- epilogue: scale / shift / LeakyReLU / shortcut (bf16) ----
    sc = (_rand(cout, seed=4).abs() + 0.5).to(dev); sh = _rand(cout, seed=5).to(dev)
    res = _bf(_rand(n, ho, wo, cout, seed=6)).to(dev)
    o, _ = ops.conv2d_fwd_b16(x, wt.reshape(-1), cout, k, st, sc, sh, ops.ACT_LEAKY, 0.1, residual=res)
    t = ref_y * sc.double().cpu() + sh.double().cpu()
    _ulp_close(o, torch.where(t > 0, t, 0.1 * t) + res.double().cpu(), "fwd epilogue")
    # ---- data gradient (transposed bank [Cin][T][Cout]) ----
    wt_t = wt.reshape(cout, T, cin).permute(2, 1, 0).contiguous().reshape(-1)
    dx = ops.conv2d_bwd_data_b16(dy, wt_t, (h, w), cin, k, st)
    ref_dx = xd.grad.permute(0, 2, 3, 1)
    _ulp_close(dx, ref_dx, "dgrad")
    base = _bf(_rand(n, h, w, cin, seed=7)).to(dev)
    acc = base.clone()
    ops.conv2d_bwd_data_b16(dy, wt_t, (h, w), cin, k, st, out=acc, accumulate=True)
    _ulp_close(acc, ref_dx + base.double().cpu(), "dgrad accumulate")
    lib().set_tuning(b"Db16", 0)               # (the register-bank stride-2 kernel off: the gathered parity classes)
    try:
        _ulp_close(ops.conv2d_bwd_data_b16(dy, wt_t, (h, w), cin, k, st), ref_dx, "dgrad, gathered classes")
    finally:
        lib().set_tuning(b"Db16", 1)
    assert torch.equal(ops.conv2d_bwd_data_b16(dy, wt_t, (h, w), cin, k, st), dx)      # bitwise repeatable
    dx32 = ops.conv2d_bwd_data_b16(dy, wt_t, (h, w), cin, k, st, out_f32=True)
    assert float((dx32.double().cpu() - ref_dx).abs().max()) <= 3e-5 * max(1.0, float(ref_dx.abs().max()))
    # ---- weight gradient: fp32 out, fp32 accumulation of exact bf16 products ----
    from dcnet_amd.lib import lib
    ref_dw = wd.grad.permute(0, 2, 3, 1)
    try:
        for knob in (0, 1):                    # the per-tap tile (default) and the filter-row kernel on bf16 inputs (where its shape test admits the layer)
            lib().set_tuning(b"w3b16", knob)
            dw = ops.conv2d_bwd_weight_b16(x, dy, k, st)
            assert dw.dtype == torch.float32 and dw.shape == (cout, k, k, cin)
            assert float((dw.double().cpu() - ref_dw).abs().max()) <= 5e-5 * max(1.0, float(ref_dw.abs().max())), knob
            assert torch.equal(ops.conv2d_bwd_weight_b16(x, dy, k, st), dw)
        lib().set_tuning(b"w3b16", 0)
        lib().set_tuning(b"9b16", 0)           # (the nine-tap kernel off: the per-tap tile for the narrow layers too)
        dw0 = ops.conv2d_bwd_weight_b16(x, dy, k, st)
        assert float((dw0.double().cpu() - ref_dw).abs().max()) <= 5e-5 * max(1.0, float(ref_dw.abs().max()))
    finally:
        lib().set_tuning(b"w3b16", 0)
        lib().set_tuning(b"9b16", 1)


@pytest.mark.parametrize("shape", [(2, 26, 26, 128, 256, 3), (3, 13, 13, 256, 128, 1), (1, 20, 12, 64, 64, 3)])
def test_b16_dgrad_batchnorm_tap_equals_the_reduce_pass(shape):
    """A stride-1 data gradient that completes the gradient of a conv + BatchNorm output forms that BatchNorm's backward partial sums in
    its epilogue: they must equal dcn_bn_act_bwd_reduce_b16 on the gradient as stored."""
    from dcnet_amd import ops
    dev = torch.device("cuda:0")
    n, h, w, cout, cin, k = shape           # the convolution maps cin -> cout; its data gradient has cin "filters"
    T = k * k
    dy = _bf(_rand(n, h, w, cout, seed=1) / 4).to(dev)
    wt_t = _bf(_rand(cin, T, cout, seed=2) / (cout * T) ** 0.5).to(dev).reshape(-1)
    yprev = _bf(_rand(n, h, w, cin, seed=3)).to(dev)
    mean = _rand(cin, seed=4, scale=0.1).to(dev); invstd = (_rand(cin, seed=5).abs() + 0.5).to(dev)
    gamma = _rand(cin, seed=6).to(dev); beta = _rand(cin, seed=7, scale=0.3).to(dev)
    tap = dict(y=yprev, mean=mean, invstd=invstd, gamma=gamma, beta=beta, act=ops.ACT_LEAKY, slope=0.1)
    dx, part = ops.conv2d_bwd_data_b16(dy, wt_t, (h, w), cin, k, 1, tap=tap)
    assert part is not None
    plain = ops.conv2d_bwd_data_b16(dy, wt_t, (h, w), cin, k, 1)
    assert torch.equal(dx, plain)
    ref_part, r = ops._bn_bwd_partials(yprev, dx, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1, None)
    a = part.double().sum(0).cpu(); b = ref_part[:r * 2 * cin].reshape(r, 2, cin).double().sum(0).cpu()
    assert torch.allclose(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()))


def test_b16_batchnorm_passes_match_their_formulas():
    from dcnet_amd import ops
    dev = torch.device("cuda:0")
    for (rows_shape, c, y_f32) in (((2, 13, 13), 128, False), ((1, 37, 5), 32, True), ((3, 8, 8), 264, False)):
        n_rows = rows_shape[0] * rows_shape[1] * rows_shape[2]
        y = _rand(*rows_shape, c, seed=1)
        y = (y if y_f32 else _bf(y)).to(dev)
        scale = (_rand(c, seed=2).abs() + 0.3).to(dev); shift = _rand(c, seed=3).to(dev)
        res = _bf(_rand(*rows_shape, c, seed=4)).to(dev)
        out = ops.scale_act(y, scale, shift, ops.ACT_LEAKY, 0.1, residual=res, out_b16=True)
        t = y.double().cpu() * scale.double().cpu() + shift.double().cpu()
        _ulp_close(out, torch.where(t > 0, t, 0.1 * t) + res.double().cpu(), "scale_act")
        # into a channel slice of a wider buffer
        wide = torch.zeros(*rows_shape, c + 64, dtype=torch.bfloat16, device=dev)
        ops.scale_act(y, scale, shift, ops.ACT_NONE, 0.0, out=wide[..., 64:])
        _ulp_close(wide[..., 64:], t, "scale_act slice")
        assert float(wide[..., :64].abs().max()) == 0.0
        # backward
        mean = _rand(c, seed=5, scale=0.2).to(dev); invstd = (_rand(c, seed=6).abs() + 0.5).to(dev)
        gamma = _rand(c, seed=7).to(dev); beta = _rand(c, seed=8, scale=0.3).to(dev)
        dout = _bf(_rand(*rows_shape, c, seed=9)).to(dev)
        dy, dgamma, dbeta = ops.bn_act_bwd(y, dout, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1)
        assert dy.dtype == torch.bfloat16
        yd, dd = y.double().cpu().reshape(-1, c), dout.double().cpu().reshape(-1, c)
        xh = (yd - mean.double().cpu()) * invstd.double().cpu()
        g = torch.where(gamma.double().cpu() * xh + beta.double().cpu() <= 0, 0.1 * dd, dd)
        sg, sgx = g.sum(0), (g * xh).sum(0)
        ref = gamma.double().cpu() * invstd.double().cpu() * (g - sg / n_rows - xh * sgx / n_rows)
        _ulp_close(dy.reshape(-1, c), ref, "bn backward", ulps=1.5, floor=2e-6)
        assert torch.allclose(dbeta.double().cpu(), sg, rtol=1e-4, atol=1e-4 * float(sg.abs().max()))
        assert torch.allclose(dgamma.double().cpu(), sgx, rtol=1e-4, atol=1e-4 * float(sgx.abs().max()))


def test_b16_layout_kernels():
    from dcnet_amd import ops
    dev = torch.device("cuda:0")
    up = _bf(_rand(2, 5, 7, 64, seed=1)).to(dev); lat = _bf(_rand(2, 10, 14, 96, seed=2)).to(dev)
    buf = torch.zeros(2, 10, 14, 160, dtype=torch.bfloat16, device=dev)
    ops.upsample2_into(up, buf[..., :64]); ops.copy_slice(lat, buf[..., 64:])
    ref = torch.cat([up.repeat_interleave(2, 1).repeat_interleave(2, 2), lat], dim=3)
    assert torch.equal(buf, ref)
    d = _bf(_rand(2, 10, 14, 160, seed=3)).to(dev)
    dsrc = torch.empty(2, 5, 7, 64, dtype=torch.bfloat16, device=dev)
    ops.upsample2_bwd(d[..., :64], dsrc, False)
    s4 = d[..., :64].double().reshape(2, 5, 2, 7, 2, 64).sum((2, 4))
    _ulp_close(dsrc, s4, "upsample2 bwd")
    ops.upsample2_bwd(d[..., :64], dsrc, True)
    _ulp_close(dsrc, s4 * 2, "upsample2 bwd accumulate", ulps=2.0)
    acc = lat.clone()
    ops.copy_slice(d[..., 64:], acc, accumulate=True)
    _ulp_close(acc, lat.double() + d[..., 64:].double(), "copy_slice accumulate")
    x = _rand(3, 4, 4, 72, seed=4).to(dev)
    assert torch.equal(ops.to_b16(x), x.to(torch.bfloat16)) and torch.equal(ops.to_f32(ops.to_b16(x)), x.to(torch.bfloat16).float())


def test_b16_backbone_training_step_against_the_fp32_run():
    """The whole backbone in bf16 storage, forward + backward in train mode: taps and parameter gradients stay within what bf16 rounding
    over ~75 layers does to this random-init network (the bf16-operand mode's own distance from fp32 is the yardstick), nothing is
    non-finite, fp32 is what comes out at the boundary, and the step is bitwise repeatable."""
    from dcnet_amd import ops
    from dcnet_amd.darknet import Darknet
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = Darknet(config_path="", img_size=128).to(dev).train()
    img = _rand(4, 3, 128, 128, seed=1).to(dev)
    gouts = None
    res = {}
    for mode in ("fp32", "bf16", "bf16s", "bf16s"):
        ops.set_precision(mode)
        try:
            for p in net.parameters():
                p.grad = None
            taps = net.forward_nhwc(img)
            if gouts is None:
                gouts = [_rand(*t.shape, seed=10 + i).to(dev) / t[0].numel() ** 0.5 for i, t in enumerate(taps)]
            torch.autograd.backward(taps, gouts)
            torch.cuda.synchronize()
        finally:
            ops.set_precision("fp32")
        grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        key = mode if mode not in res else mode + "_again"
        res[key] = ([t.detach().clone() for t in taps], grads)
    assert all(t.dtype == torch.float32 and bool(torch.isfinite(t).all()) for t in res["bf16s"][0])
    assert all(bool(torch.isfinite(g).all()) for g in res["bf16s"][1].values())
    assert set(res["bf16s"][1]) == set(res["fp32"][1])
    for a, b in zip(res["bf16s"][0], res["bf16s_again"][0]):
        assert torch.equal(a, b)
    for k_ in res["bf16s"][1]:
        assert torch.equal(res["bf16s"][1][k_], res["bf16s_again"][1][k_]), k_

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-20))
    for i in range(3):
        d_ops, d_sto = rel(res["bf16"][0][i], res["fp32"][0][i]), rel(res["bf16s"][0][i], res["fp32"][0][i])
        assert d_sto < max(4 * d_ops, 0.05), (i, d_ops, d_sto)
    worst = 0.0
    for k_, gref in res["fp32"][1].items():
        if gref.numel() < 64 or float(gref.norm()) == 0.0:
            continue
        d_ops, d_sto = rel(res["bf16"][1][k_], gref), rel(res["bf16s"][1][k_], gref)
        worst = max(worst, d_sto)
        assert d_sto < max(5 * d_ops, 0.25), (k_, d_ops, d_sto)
    assert worst > 1e-4          # it IS reduced precision


def test_coattention_on_one_f16_piece_in_the_bf16_modes():
    """In the bf16 precision modes the nine co-attention products per scale run on gemm3.hip with ONE f16 piece per operand (the high piece of
    the split form: 11 significant bits, one MFMA per product).  Forward and backward stay within f16-operand rounding of the fp32-accurate
    run (two pieces), differ from it (the mode is live), and the knob `H1gemm3=0` gives the two-piece result back bit for bit."""
    from dcnet_amd import ops
    from dcnet_amd.functions import CoAttentionPairs
    from dcnet_amd.lib import lib
    dev = torch.device("cuda:0")
    n, h, w, c = 4, 26, 26, 512
    fv0 = torch.nn.functional.normalize(_rand(n, h, w, c, seed=1), dim=3).to(dev)
    g = (_rand(n, h, w, 2 * c, seed=2) / (h * w) ** 0.5).to(dev)
    res = {}
    try:
        for tag, mode, knob in (("fp32", "fp32", 1), ("h1", "bf16s", 1), ("two", "bf16s", 0)):
            ops.set_precision(mode); lib().set_tuning(b"H1gemm3", knob)
            fv = fv0.clone().requires_grad_(True)
            out = CoAttentionPairs.apply(fv, 10.0)
            out.backward(g)
            res[tag] = (out.detach().clone(), fv.grad.clone())
    finally:
        ops.set_precision("fp32"); lib().set_tuning(b"H1gemm3", 1)
    assert lib().gemm3_supported(h * w, c, h * w, n // 2) == 1
    for k_ in (0, 1):
        assert torch.equal(res["two"][k_], res["fp32"][k_])                          # the two-piece path is the fp32 one
        ref = res["fp32"][k_].double(); got = res["h1"][k_].double()
        err = float((got - ref).abs().max()); scale = float(ref.abs().max())
        assert 1e-7 * scale < err < 4e-3 * scale, (k_, err, scale)                   # live, and within f16-operand rounding (2^-11 per operand)
        cos = float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0))
        assert cos > 0.99999, (k_, cos)


def test_b16_inference_models_run_on_bf16_storage():
    """Eval mode on bf16 storage — the pair model and the n_frame model (folded BatchNorm in the conv epilogues, bf16 taps into the head, fp32
    outputs): finite, close to the fp32 run by the mode's own measure, and the top-k candidate cache takes its outputs as they come."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from util import build_product, maxdiff, synth_sd
    from dcnet_amd import ops, postprocess as PP
    from dcnet_amd.utils.synth import synth_inputs
    dev = torch.device("cuda:0")
    size, b, t = 256, 2, 3
    sd = synth_sd(size)
    image, word_id, word_mask = synth_inputs(b * t, size, n_queries=b, seed=5)
    res = {}
    try:
        for mode in ("fp32", "bf16s"):
            ops.set_precision(mode)
            m = build_product(size, sd, dev, test_model=True).eval()
            with torch.no_grad():
                res[mode] = m(image.to(dev), word_id.to(dev), word_mask.to(dev), t)
            if mode == "bf16s":
                outbox, sim, loc, corr, only_obj = res[mode]
                one = torch.ones(b, device=dev); zero = torch.zeros(b, device=dev)
                boxes, score, feat, cells = PP.topk_candidates(list(outbox), list(corr), size, 4, one, zero, zero, torch.full((b, 2), size, device=dev))
                assert bool(torch.isfinite(boxes).all()) and bool(torch.all(score[:, :-1] >= score[:, 1:]))
    finally:
        ops.set_precision("fp32")
    for s_ in range(3):
        a, r = res["bf16s"][0][s_], res["fp32"][0][s_]
        assert a.dtype == torch.float32 and bool(torch.isfinite(a).all())
        d = maxdiff(a, r)
        assert 1e-4 < d < 0.5 * float(r.abs().max()), (s_, d)


def test_stem_backward_reads_a_bf16_gradient_as_it_is():
    """bf16 storage: the gradient that reaches the 3-channel stem is a bf16 tensor (the data gradient of the layer behind it), its raw output y
    stays fp32.  The fused stem backward (BatchNorm + LeakyReLU backward + weight gradient in one pass) reads the bf16 gradient directly:
    bitwise what it computes from the same values cast to fp32 first."""
    from dcnet_amd import ops
    dev = torch.device("cuda:0")
    n, h, w = 2, 40, 48
    x = _rand(n, h, w, 4, seed=1).to(dev); x[..., 3] = 0
    y = _rand(n, h, w, 32, seed=2).to(dev)
    dout16 = _bf(_rand(n, h, w, 32, seed=3)).to(dev)
    mean = _rand(32, seed=4, scale=0.1).to(dev); invstd = (_rand(32, seed=5).abs() + 0.5).to(dev)
    gamma = _rand(32, seed=6).to(dev); beta = _rand(32, seed=7, scale=0.3).to(dev)
    a = ops.stem_bwd_weight_bn(x, y, dout16, mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1)
    b = ops.stem_bwd_weight_bn(x, y, dout16.float(), mean, invstd, gamma, beta, ops.ACT_LEAKY, 0.1)
    assert torch.equal(a[0], b[0])                                           # the weight gradient: same values, same order
    for u, v in zip(a[1:], b[1:]):                                           # d gamma, d beta: the reduce passes differ in their block shape only
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-5 * float(v.abs().max()))
