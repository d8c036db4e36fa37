"""Micro-benchmark of csrc/gemm3.hip at the co-attention shapes of the C2 workload (32 pairs, 512 channels):
NT affinity (hw x hw x c), NN / TN attended products (hw x c x hw).  Usage (on the GPU box): python tools/bench_gemm3.py [--g 52]"""
import argparse
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
    ap = argparse.ArgumentParser(); ap.add_argument("--g", type=int, default=52); ap.add_argument("--b", type=int, default=32)
    ap.add_argument("--c", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--abl", type=str, default="0", help="timing-only ablations to run (bits: 1 no DMA, 2 no fragment reads, 4 no MFMAs; needs G3_ABL 1) + 16 * schedule variant")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hw, b, c = args.g * args.g, args.b, args.c
    ld = (hw + 31) // 32 * 32
    f1 = torch.nn.functional.normalize(torch.randn(b, hw, c, device=dev), dim=2)
    f2 = torch.nn.functional.normalize(torch.randn(b, hw, c, device=dev), dim=2)
    E = torch.zeros(b, hw, ld, device=dev); E[:, :, :hw] = torch.rand(b, hw, hw, device=dev)
    one = ops.absmax(torch.ones(8, device=dev))
    f1s, f2s = ops.gemm3_presplit(f1, one), ops.gemm3_presplit(f2, one)
    Es = ops.gemm3_presplit(E, one)
    out_e = torch.empty(b, hw, ld, device=dev)
    out_f = torch.empty(b, hw, c, device=dev)
    rows = [("presplit f (hw x c)", lambda: ops.gemm3_presplit(f1, one, out=f1s), 0.0, 2.0 * b * hw * c * 4),
            ("presplit E (hw x hw)", lambda: ops.gemm3_presplit(E, one, out=Es), 0.0, 2.0 * b * hw * ld * 4),
            ("NT  f1 . f2^T", lambda: ops.gemm3(f1s, f2s, out_e, hw, hw, c, one, one), 2.0 * b * hw * hw * c, 0.0),
            ("NN  E . f2", lambda: ops.gemm3(Es, f2s, out_f, hw, c, hw, one, one, b_t=True), 2.0 * b * hw * hw * c, 0.0),
            ("TN  E^T . f1", lambda: ops.gemm3(Es, f1s, out_f, hw, c, hw, one, one, a_t=True, b_t=True), 2.0 * b * hw * hw * c, 0.0)]
    from dcnet_amd.lib import lib
    for abl in [int(v) for v in args.abl.split(",")]:
      lib().set_tuning(b"Gemm3", 1 + 16 * (abl & 15) + 256 * (abl >> 4))
      print("ablation", abl)
      for name, fn, flop, byt in rows:
        if abl and not flop:
            continue
        ms = timeit(fn, args.iters)
        print(f"{name:24s} {ms:8.3f} ms" + (f"  {flop / ms / 1e9:7.1f} TFLOP/s  {flop / ms / 1e9 / 838.9:5.3f} of 838.9" if flop else f"  {byt / ms / 1e9:7.2f} TB/s"))
    lib().set_tuning(b"Gemm3", 1)
    # the engines these products ran on before (dcn_coattn_fwd: NT + NN on igemm.hip, TN on wgrad.hip)
    cat = torch.empty(2, b, hw, c, device=dev)
    for knob in (0, 1):
        lib().set_tuning(b"Gemm3", knob)
        ms = timeit(lambda: ops.coattn_fwd(f1, f2, cat[0], cat[1], 10.0), 5)
        print(f"coattn_fwd  Gemm3={knob}: {ms:8.3f} ms")
    lib().set_tuning(b"Gemm3", 1)


if __name__ == "__main__":
    main()
