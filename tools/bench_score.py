"""The scoring pass (l2norm_score_fwd: corr = x / ||x||, sim = <corr, q>, neg_sim) at the three scales of the step (N = 64 maps of
the 8 x 8-frame clips, E = 512), per rows-per-wave / non-temporal setting, against a plain copy of the same bytes.
Usage (GPU box): python tools/bench_score.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcnet_amd import ops  # noqa: E402
from dcnet_amd.lib import lib  # noqa: E402


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    n, e = 64, 512
    for hw in (52, 26, 13):
        x = torch.randn(n, hw, hw, e, device=dev)
        # a second set of buffers so consecutive launches do not hit the same lines in the memory-side cache
        xs = [torch.randn_like(x) for _ in range(4)]
        q = torch.randn(n, e, device=dev)
        outs = [torch.empty_like(x) for _ in range(4)]
        by = 2 * x.numel() * 4
        i = [0]

        def copy():
            i[0] = (i[0] + 1) & 3
            outs[i[0]].copy_(xs[i[0]])
        t_c = timeit(copy)
        line = "%2dx%2d copy %.4f ms %.2f TB/s |" % (hw, hw, t_c, by / t_c / 1e9)
        ref = None
        for nt in (0, 1):
            for rpw in (1, 2, 4, 8):
                lib().set_tuning(b"e2rpw", rpw); lib().set_tuning(b"f2nt", nt)

                def run():
                    i[0] = (i[0] + 1) & 3
                    return ops.l2norm_score_fwd(xs[i[0]], q, hw * hw, out=outs[i[0]], want_flip=True)
                t = timeit(run)
                o = ops.l2norm_score_fwd(x, q, hw * hw, want_flip=True)
                if ref is None:
                    ref = o
                else:
                    assert all(torch.equal(a, b) for a, b in zip(o, ref)), (nt, rpw)
                line += " nt%d rpw%d %.4f %.2f |" % (nt, rpw, t, by / t / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
