"""Per-layer timing of the bf16-storage convolution kernels (csrc/conv1.hip conv1b, wgrad.hip IN16) on the benchmark's layer shapes.
    python tools/bench_b16.py [--set key=value,...]   (dcn_set_tuning knobs, e.g. bwide=256, btall=512)"""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcnet_amd import ops
from dcnet_amd.lib import lib

SHAPES = [  # n, h, w, cin, cout, k, stride
    (64, 52, 52, 128, 256, 3, 1), (64, 26, 26, 256, 512, 3, 1), (64, 13, 13, 512, 1024, 3, 1), (64, 104, 104, 64, 128, 3, 1),
    (64, 52, 52, 256, 128, 1, 1), (64, 26, 26, 512, 256, 1, 1), (64, 13, 13, 1024, 512, 1, 1),
    (64, 52, 52, 512, 512, 3, 1), (64, 52, 52, 1024, 512, 1, 1), (64, 52, 52, 512, 512, 1, 1),
    (64, 104, 104, 128, 256, 3, 2), (64, 208, 208, 32, 64, 3, 1), (64, 416, 416, 32, 64, 3, 2),
]


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--set", default=""); ap.add_argument("--ab", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    variants = [("base", args.set)] + ([("ab", args.ab)] if args.ab else [])
    rows = []
    for (n, h, w, cin, cout, k, st) in SHAPES:
        x = torch.randn(n, h, w, cin, device=dev).bfloat16()
        wt = (torch.randn(cout, k, k, cin, device=dev) / (cin * k * k) ** 0.5).bfloat16().reshape(-1)
        ho, wo = ops.conv_out_hw(h, w, k, st)
        dy = torch.randn(n, ho, wo, cout, device=dev).bfloat16()
        wtt = wt.reshape(cout, k * k, cin).permute(2, 1, 0).contiguous().reshape(-1)
        gf = 2.0 * n * ho * wo * cout * cin * k * k / 1e9
        line = f"{cin:5d}->{cout:4d} k{k} s{st} @{h:3d}  {gf:7.1f} GF |"
        for name, knobs in variants:
            for kv in [t for t in knobs.split(",") if t]:
                k_, v_ = kv.split("="); lib().set_tuning(k_.encode(), int(v_))
            tf = timeit(lambda: ops.conv2d_fwd_b16(x, wt, cout, k, st, want_stats=True))
            td = timeit(lambda: ops.conv2d_bwd_data_b16(dy, wtt, (h, w), cin, k, st))
            tw = timeit(lambda: ops.conv2d_bwd_weight_b16(x, dy, k, st))
            line += f" {name}: fwd {tf:.3f} ({gf / tf:6.0f}) dgrad {td:.3f} ({gf / td:6.0f}) wgrad {tw:.3f} ({gf / tw:6.0f}) |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
