import sys, os, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_f8_gpu import _mx_rows, CASES
from test_b16_gpu import _bf, _rand
from dcnet_amd import ops
dev = torch.device("cuda:0")
for case in CASES[:6]:
    n, h, w, cin, cout, k, st = case
    T = k * k
    for flat in (0, 1):
        mag = torch.logspace(-2, 1, n * h * w)[torch.randperm(n * h * w, generator=torch.Generator().manual_seed(5))].reshape(n, h, w, 1)
        x = _bf(_rand(n, h, w, cin, seed=1) * (1.0 if flat else mag))
        wt = _bf(_rand(cout, k, k, cin, seed=2) / (cin * T) ** 0.5 * (1.0 if flat else torch.logspace(-1, 1, cout).reshape(cout, 1, 1, 1)))
        x8, xs = ops.quant_rows_e4m3(x.to(dev)); w8, ws = ops.quant_rows_e4m3(wt.reshape(cout, T * cin).to(dev))
        _, _, xd = _mx_rows(x); _, _, wd = _mx_rows(wt.reshape(cout, T * cin))
        ref = F.conv2d(xd.permute(0, 3, 1, 2), wd.reshape(cout, k, k, cin).permute(0, 3, 1, 2), stride=st, padding=(k - 1) // 2).permute(0, 2, 3, 1)
        y32, _ = ops.conv2d_fwd_f8(x8, xs, w8.reshape(-1), ws, cout, k, st, out_f32=True)
        err = (y32.double().cpu() - ref).abs()
        # error relative to the sum of |products| of each output (what an fp32 accumulation is measured against)
        refabs = F.conv2d(xd.abs().permute(0, 3, 1, 2), wd.abs().reshape(cout, k, k, cin).permute(0, 3, 1, 2), stride=st, padding=(k - 1) // 2).permute(0, 2, 3, 1)
        print(case, "flat" if flat else "scaled", "max err/max|y| %.2e  max err/sum|prod| %.2e  median %.2e" % (
            float(err.max() / ref.abs().max()), float((err / refabs.clamp_min(1e-30)).max()), float((err / refabs.clamp_min(1e-30)).median())), flush=True)
