"""Probe: what would a 256 x 256 weight-gradient tile buy the 1x1 layers?  The TN form of csrc/gemm3.hip (256 x 256 tile, eight waves, both
operands by LDS-DMA) computes dW = dY^T X per K-chunk with the chunk index as the batch (the slabs a split-K weight gradient writes), on
operands that were split BEFORE the timed region — an upper bound for a weight-gradient kernel on that tile, which would have to split its
operands while it stages them.  Printed beside the product's weight gradient (dcn_conv2d_bwd_weight) of the same layer.
Usage (GPU box): python tools/wgrad_tile_probe.py"""
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
    for cin, cout, n, h in ((1024, 512, 64, 52), (512, 512, 64, 52), (512, 256, 64, 52), (256, 512, 64, 52), (256, 128, 64, 52), (512, 256, 64, 26), (1024, 512, 64, 13)):
        m = n * h * h
        tiles = ((cout + 255) // 256) * ((cin + 255) // 256)
        splits = max(1, 256 // tiles)
        kchunk = (m // splits) // 16 * 16
        mm = kchunk * splits                          # (the tail pixels are dropped: timing only)
        x = torch.randn(splits, kchunk, cin, device=dev)
        dy = torch.randn(splits, kchunk, cout, device=dev) / 8
        ax, ady = ops.absmax(x), ops.absmax(dy)
        xs, dys = ops.gemm3_presplit(x, ax), ops.gemm3_presplit(dy, ady)
        slabs = torch.empty(splits, cout, cin, device=dev)
        flop = 2.0 * mm * cin * cout
        t_g = timeit(lambda: ops.gemm3(dys, xs, slabs, cout, cin, kchunk, ady, ax, a_t=True, b_t=True))
        t_p = timeit(lambda: (ops.gemm3_presplit(x, ax, out=xs), ops.gemm3_presplit(dy, ady, out=dys)))
        x4 = x.view(n, -1, cin)[:, : h * h].reshape(n, h, h, cin) if mm == m else torch.randn(n, h, h, cin, device=dev)
        dy4 = torch.randn(n, h, h, cout, device=dev) / 8
        a4x, a4d = ops.absmax(x4), ops.absmax(dy4)
        t_w = timeit(lambda: ops.conv2d_bwd_weight(x4, dy4, 1, 1, amax_x=a4x, amax_dy=a4d))
        print(f"{cin:5d} -> {cout:4d} @{h:3d}: gemm3 TN {splits:3d} x K {kchunk:6d}: {t_g:.3f} ms ({flop / t_g / 1e9:6.1f} TFLOP/s = {flop / t_g / 1e9 / 838.9:.3f}); "
              f"split passes {t_p:.3f} ms; product's weight gradient {t_w:.3f} ms ({2.0 * m * cin * cout / t_w / 1e9:6.1f})")


if __name__ == "__main__":
    main()
