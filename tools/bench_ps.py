"""Experiment: what a pre-split activation operand is worth to conv1.hip (1x1 / stride-2 forward and data gradient).
x is split once (gemm3_presplit) and the kernel's PS build reads the pieces instead of splitting fragments in registers:
    python tools/bench_ps.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcnet_amd import ops  # noqa: E402
from dcnet_amd.lib import lib  # noqa: E402


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
    n = 64
    shapes = [(1024, 512, 1, 1, 52), (512, 512, 1, 1, 52), (256, 512, 1, 1, 52), (256, 128, 1, 1, 52), (128, 64, 1, 1, 104), (512, 256, 1, 1, 26),
              (1024, 512, 1, 1, 13), (128, 256, 3, 2, 104), (256, 512, 3, 2, 52)]
    for cin, cout, k, st, h in shapes:
        x = torch.randn(n, h, h, cin, device=dev)
        w = torch.randn(cout, k, k, cin, device=dev) * 0.05
        ax, aw = ops.absmax(x), ops.absmax(w)
        xs = ops.gemm3_presplit(x.view(1, -1, cin), ax).view(n, h, h, cin)
        f0 = lambda: ops.conv2d_fwd(x, w, k, st, want_stats=True, amax_x=ax, amax_w=aw)
        f1 = lambda: ops.conv2d_fwd(xs, w, k, st, want_stats=True, amax_x=ax, amax_w=aw)
        y0, s0 = f0()
        lib().set_tuning(b"1ps", 1)
        y1, s1 = f1()
        lib().set_tuning(b"1ps", 0)
        same = bool(torch.equal(y0, y1)) and bool(torch.equal(s0, s1))
        best = [1e9, 1e9]
        for _ in range(3):
            for arm, fn in ((0, f0), (1, f1)):
                lib().set_tuning(b"1ps", arm)
                best[arm] = min(best[arm], timeit(fn))
        lib().set_tuning(b"1ps", 0)
        gf = 2.0 * n * (h // st) ** 2 * cout * k * k * cin / 1e9
        print(f"fwd {cin}->{cout} k{k} s{st} @{h}: in-register split {best[0]:.4f} ms ({gf / best[0]:.0f} TF/s)  pre-split {best[1]:.4f} ms "
              f"({gf / best[1]:.0f} TF/s)  bitwise equal: {same}", flush=True)


if __name__ == "__main__":
    main()
