"""What the reduced-precision modes deliver end to end at the geometry of tests/test_model_gpu.py::test_reduced_precision_modes_end_to_end
(256 x 256, 4 images, synthetic weights): outbox difference from fp32 relative to max|outbox| per scale, relative loss-term differences,
cosine of the bbox head's weight gradient.  The test's limits are these figures + 20 % head-room.  Usage: python tools/measure_reduced_modes.py"""
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_product, maxdiff, synth_sd  # noqa: E402


def run(mode, dev, size=256, n=4, seed=11):
    from dcnet_amd import losses, ops
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs
    sd = synth_sd(size)
    image, word_id, word_mask = synth_inputs(n, size, seed=seed)
    bbox = synth_boxes(n, size, seed=seed).to(dev)
    ops.set_precision(mode)
    try:
        m = build_product(size, sd, dev).eval()
        with torch.no_grad():
            outbox = m(image.to(dev), word_id.to(dev), word_mask.to(dev))[0]
        m.train()
        random.seed(13)
        out = m(image.to(dev), word_id.to(dev), word_mask.to(dev))
        loss, parts = losses.total_loss(out, bbox, size)
        loss.backward()
        g = m.fcn_out[0][1].weight.grad
        return [o.detach().cpu() for o in outbox], {k: float(v.detach()) for k, v in parts.items()}, g.detach().cpu()
    finally:
        ops.set_precision("fp32")


def main():
    dev = torch.device("cuda:0")
    ref = run("fp32", dev)
    out = {}
    for mode in ("bf16", "bf16s", "fp8", "fp8s"):
        r = run(mode, dev)
        box = [maxdiff(a, b) / float(a.abs().max()) for a, b in zip(ref[0], r[0])]
        lossd = {k: abs(r[1][k] - v) / max(abs(v), 1e-3) for k, v in ref[1].items()}
        cos = float(torch.nn.functional.cosine_similarity(ref[2].flatten().double(), r[2].flatten().double(), dim=0))
        out[mode] = {"box_rel": [round(b, 4) for b in box], "loss_rel": {k: round(v, 4) for k, v in lossd.items()}, "grad_cos": round(cos, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
