"""Per-layer micro-benchmark of the conv engine at the C2 workload (N=64 images of 416x416):
forward, data gradient and weight gradient of every distinct conv shape of the backbone and head.
Usage (on the GPU box): python tools/bench_convs.py [--n 64] [--size 416]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcnet_amd import darknet as D, ops  # noqa: E402


def shapes(size):
    d = D.Darknet(config_path="")
    table = {}
    res = {-1: size}
    out = collections.Counter()
    for op in d._plan:
        if isinstance(op, D._ConvOp):
            h = res[op.src]
            res[op.dst] = h // op.stride
            out[(op.cin if op.cin > 4 else 4, op.cout, op.k, op.stride, h)] += 1
        elif isinstance(op, D._UpCatOp):
            res[op.dst] = res[op.lat_src]
        else:
            res[op.dst] = res[op.src]
    for s, g in enumerate((size // 32, size // 16, size // 8)):
        cin = (1024, 512, 256)[s]
        out[(cin, 512, 1, 1, g)] += 1        # mapping_visu
        out[(1024, 512, 1, 1, g)] += 1       # corr_conv
        out[(1056, 512, 1, 1, g)] += 1       # fcn_emb.0 (1032 padded)
        out[(512, 512, 3, 1, g)] += 1
        out[(512, 512, 1, 1, g)] += 1
        out[(512, 256, 1, 1, g)] += 1
        out[(256, 32, 1, 1, g)] += 1         # bbox head (15 padded to 32)
    return out


ITERS = 5


def timeit(fn, iters=None):
    iters = ITERS if iters is None else iters
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=64); ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--only", type=str, default="", help="cin,cout,k,stride,h: benchmark a single shape")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--set", type=str, default="", help="key=value[,key=value] tuning knobs applied to both arms")
    ap.add_argument("--ab-default", type=int, default=0, help="the knob's default value (restored for the A arm)")
    ap.add_argument("--ab", type=str, default="", help="key=value tuning knob (dcn_set_tuning) measured against the default, same process")
    ap.add_argument("--strip", action="store_true", help="only the 3x3 stride-1 layers with more than 64 filters (the strip kernels' launches)")
    ap.add_argument("--k1", action="store_true", help="only the 1x1 layers and the stride-2 3x3 layers (conv1.hip's launches)")
    ap.add_argument("--no-wgrad", action="store_true", help="skip the weight gradients")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.set:
        from dcnet_amd.lib import lib as _lib
        for kv in args.set.split(","):
            kk, vv = kv.split("=")
            _lib().set_tuning(kk.encode(), int(vv))
    tot = collections.Counter()
    rows = []
    table = shapes(args.size)
    if args.only:
        key = tuple(int(v) for v in args.only.split(","))
        table = {key: table.get(key, 1)}
    if args.strip:
        table = {k_: v_ for k_, v_ in table.items() if k_[2] == 3 and k_[3] == 1 and k_[1] > 64 and k_[0] >= 32}
    if args.k1:
        table = {k_: v_ for k_, v_ in table.items() if (k_[2] == 1 or k_[3] == 2) and k_[0] >= 32}
    global ITERS
    ITERS = args.iters
    for (cin, cout, k, st, h), cnt in sorted(table.items(), key=lambda kv: -kv[0][4]):
        x = torch.randn(args.n, h, h, cin, device=dev)
        w = torch.randn((cout, 64) if cin == 4 else (cout, k, k, cin), device=dev) * 0.05
        ho = h // st
        dy = torch.randn(args.n, ho, ho, cout, device=dev)
        flop = 2.0 * args.n * ho * ho * cout * k * k * (3 if cin == 4 else cin)
        # abs-max words of the operands, computed once (in the model the producing kernels maintain them)
        ax, aw, ady = (ops.absmax(t_) if ops.use_amax() else None for t_ in (x, w, dy))
        f_fwd = lambda: ops.conv2d_fwd(x, w, k, st, want_stats=True, amax_x=ax, amax_w=aw)
        f_dg = lambda: ops.conv2d_bwd_data(dy, w, (h, h), k, st, amax_dy=ady, amax_w=aw)
        has_dg = cin != 4 and cout % 32 == 0
        t_f = timeit(f_fwd)
        t_w = 1e-9 if args.no_wgrad else timeit(lambda: ops.conv2d_bwd_weight(x, dy, k, st, amax_x=ax, amax_dy=ady))
        t_d = timeit(f_dg) if has_dg else 0.0
        if args.ab:
            # interleaved rounds in one process (default, knob, default, knob, ...), minimum per arm
            from dcnet_amd.lib import lib
            key, val = args.ab.split("=")
            f_w = lambda: ops.conv2d_bwd_weight(x, dy, k, st, amax_x=ax, amax_dy=ady)
            arms = {0: [[], [], []], 1: [[], [], []]}
            for rnd in range(4):
                for arm in (0, 1):
                    lib().set_tuning(key.encode(), int(val) if arm else args.ab_default)
                    arms[arm][0].append(timeit(f_fwd)); arms[arm][1].append(timeit(f_dg) if has_dg else 0.0)
                    arms[arm][2].append(timeit(f_w))
            lib().set_tuning(key.encode(), args.ab_default)
            t_f, t_d, t_w = (min(v) for v in arms[0])
            t_f2, t_d2, t_w2 = (min(v) for v in arms[1])
            print("AB %5d %5d k%d s%d H%4d  fwd %.3f -> %.3f  dgrad %.3f -> %.3f  wgrad %.3f -> %.3f" % (cin, cout, k, st, h, t_f, t_f2, t_d, t_d2, t_w, t_w2))
            tot["fwd_b"] += cnt * t_f2; tot["dgrad_b"] += cnt * t_d2; tot["wgrad_b"] += cnt * t_w2
        rows.append((cin, cout, k, st, h, cnt, flop / 1e9, t_f, t_d, t_w))
        tot["fwd"] += cnt * t_f; tot["dgrad"] += cnt * t_d; tot["wgrad"] += cnt * t_w
        tot["flop"] += cnt * flop
    print("%5s %5s k s %4s cnt %8s | %8s %6s | %8s %6s | %8s %6s" % ("cin", "cout", "H", "GF", "fwd ms", "TF/s", "dgrad ms", "TF/s", "wgrad ms", "TF/s"))
    for (cin, cout, k, st, h, cnt, gf, tf, td, tw) in rows:
        print("%5d %5d %d %d %4d %3d %8.1f | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f" % (
            cin, cout, k, st, h, cnt, gf, tf, gf / tf, td, gf / td if td else 0, tw, gf / tw))
    for k in ("fwd", "dgrad", "wgrad"):
        if not tot[k]:
            continue
        print("total %-6s %8.2f ms  -> %6.1f TF/s" % (k, tot[k], tot["flop"] / 1e9 / tot[k]))
        if args.ab:
            print("   with %-12s %8.2f ms  -> %6.1f TF/s" % (args.ab, tot[k + "_b"], tot["flop"] / 1e9 / tot[k + "_b"]))


if __name__ == "__main__":
    main()
