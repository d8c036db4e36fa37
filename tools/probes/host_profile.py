"""Where the HOST time of a training step goes (cProfile over a few steps of bench.py's step; GPU box).
Usage: python tools/host_profile.py [--steps 4]"""
import argparse
import cProfile
import os
import pstats
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=4); ap.add_argument("--clips", type=int, default=8)
    args = ap.parse_args()
    from dcnet_amd import losses
    from dcnet_amd.model import grounding_model
    from dcnet_amd.parallel import freeze_gradless
    from dcnet_amd.train import make_optimizer
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = grounding_model(corpus=list(range(1000)), light=False, emb_size=512, coordmap=True, bert_model="bert-base-uncased",
                            dataset="vid", img_size=416, config_path=os.path.join(ROOT, "model", "yolov3.cfg"), weights_path=None).to(dev)
    model.train(); freeze_gradless(model)
    opt = make_optimizer(model, 1e-4)
    n = args.clips * 8
    image, word_id, word_mask = (t.to(dev) for t in synth_inputs(n, 416, seed=100))
    bbox = synth_boxes(n, 416, seed=100).to(dev)
    random.seed(13)

    def step():
        out = model(image, word_id, word_mask)
        loss, _ = losses.total_loss(out, bbox, 416)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(args.steps):
        step()
    pr.disable()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"host time per step (under cProfile): {host / args.steps * 1e3:.1f} ms")
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
    st.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
