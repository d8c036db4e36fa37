"""Streaming passes of the BatchNorm family per activation shape of the step (N = 64, 416x416), against a plain copy of the same
bytes.  Usage (GPU box): python tools/bench_bn.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcnet_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    for kv in [t for t in (sys.argv[1] if len(sys.argv) > 1 else "").split(",") if t]:       # knobs: python tools/bench_bn.py Bpc=0
        from dcnet_amd.lib import lib
        k_, v_ = kv.split("="); lib().set_tuning(k_.encode(), int(v_))
    shapes = [(416, 32, 1), (208, 64, 2), (208, 32, 1), (104, 128, 3), (104, 64, 2), (52, 256, 9), (52, 128, 8), (26, 512, 9), (26, 256, 8),
              (13, 1024, 5), (13, 512, 4), (52, 512, 6)]
    tot = {"scale_act": 0.0, "partials": 0.0, "apply": 0.0, "copy": 0.0, "copy3": 0.0}
    print("%4s %5s cnt | %9s %6s | %9s %6s | %9s %6s | copy %6s" % ("H", "C", "scale ms", "TB/s", "part ms", "TB/s", "apply ms", "TB/s", "TB/s"))
    for h, c, cnt in shapes:
        n = 64
        y = torch.randn(n, h, h, c, device=dev); res = torch.randn_like(y); dout = torch.randn_like(y)
        sc = torch.rand(c, device=dev) + 0.5; sh = torch.randn(c, device=dev)
        mean = torch.randn(c, device=dev) * 0.1; istd = torch.rand(c, device=dev) + 0.5
        am = ops.amax_slot(dev)
        out = torch.empty_like(y)
        t_s = timeit(lambda: ops.scale_act(y, sc, sh, ops.ACT_LEAKY, 0.1, residual=None, out=out, amax_out=am))
        t_sr = timeit(lambda: ops.scale_act(y, sc, sh, ops.ACT_LEAKY, 0.1, residual=res, out=out, amax_out=am))
        t_b = timeit(lambda: ops.bn_act_bwd(y, dout, mean, istd, sc, sh, ops.ACT_LEAKY, 0.1, amax_out=am))
        t_c = timeit(lambda: out.copy_(y))
        by = y.numel() * 4
        # bn_act_bwd = partials (8 B/elem) + sums + apply (12 B/elem)
        print("%4d %5d %3d | %9.4f %6.2f | (res %.4f %5.2f) | bwd %9.4f %6.2f | copy %6.2f" % (
            h, c, cnt, t_s, 2 * by / t_s / 1e9, t_sr, 3 * by / t_sr / 1e9, t_b, 5 * by / t_b / 1e9, 2 * by / t_c / 1e9))
        tot["scale_act"] += cnt * t_s; tot["apply"] += cnt * t_b; tot["copy"] += cnt * t_c
    print("sum over the step's layers: scale_act %.2f ms, bn backward (partials + sums + apply) %.2f ms; a copy of each tensor %.2f ms" % (
        tot["scale_act"], tot["apply"], tot["copy"]))


if __name__ == "__main__":
    main()
