"""Timing ablations of one conv kernel on a few layer shapes (N = 64), in one process, interleaved rounds, minimum per arm.
Needs a build with the ablation variants compiled in (e.g. DCN_EXTRA_FLAGS=-DC3_ABL=1 python -m dcnet_amd.build --force).

    python tools/bench_abl.py --knob 3abl --values 0,1,3,7,8,15,16 --shapes 128,256,3,1,52;256,512,3,1,26 [--pass fwd|dgrad|wgrad]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcnet_amd import ops  # noqa: E402
from dcnet_amd.lib import lib  # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", default="3abl"); ap.add_argument("--values", default="0,1,3,7,8,15,16")
    ap.add_argument("--shapes", default="128,256,3,1,52;256,512,3,1,26;512,1024,3,1,13;512,512,3,1,52")
    ap.add_argument("--pass", dest="which", default="fwd"); ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10); ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--set", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for kv in [t for t in a.set.split(",") if t]:
        k_, v_ = kv.split("="); lib().set_tuning(k_.encode(), int(v_))
    vals = [int(v) for v in a.values.split(",")]
    for sh in a.shapes.split(";"):
        cin, cout, k, st, h = (int(v) for v in sh.split(","))
        x = torch.randn(a.n, h, h, cin, device=dev)
        w = torch.randn(cout, k, k, cin, device=dev) * 0.05
        ho = h // st
        dy = torch.randn(a.n, ho, ho, cout, device=dev)
        ax, aw, ady = (ops.absmax(t_) for t_ in (x, w, dy))
        fn = {"fwd": lambda: ops.conv2d_fwd(x, w, k, st, want_stats=True, amax_x=ax, amax_w=aw),
              "dgrad": lambda: ops.conv2d_bwd_data(dy, w, (h, h), k, st, amax_dy=ady, amax_w=aw),
              "wgrad": lambda: ops.conv2d_bwd_weight(x, dy, k, st, amax_x=ax, amax_dy=ady)}[a.which]
        best = {v: 1e9 for v in vals}
        for _ in range(a.rounds):
            for v in vals:
                lib().set_tuning(a.knob.encode(), v)
                best[v] = min(best[v], timeit(fn, a.iters))
        lib().set_tuning(a.knob.encode(), 0)
        gf = 2.0 * a.n * ho * ho * cout * k * k * cin / 1e9
        print(f"{a.which} {cin}->{cout} k{k} s{st} @{h}: " + "  ".join(f"{a.knob}={v}: {best[v]:.3f} ms ({gf / best[v]:.0f} TF/s)" for v in vals), flush=True)


if __name__ == "__main__":
    main()
