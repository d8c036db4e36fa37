"""Host-side anatomy of a graph-replayed training step (GPU box): time in the draws, in hipGraphLaunch, and GPU time per replay.
Usage: python tools/graph_probe.py [--steps 6] [--clips 8]"""
import argparse
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=6); ap.add_argument("--clips", type=int, default=8)
    args = ap.parse_args()
    from dcnet_amd.graph import GraphedTrainStep
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
    t0 = time.perf_counter()
    step = GraphedTrainStep(model, opt, image, word_id, word_mask, bbox, 416, warmup=2)
    torch.cuda.synchronize()
    print(f"construct (2 eager steps + capture + 1 replay): {time.perf_counter() - t0:.2f} s")
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    td = tr = 0.0
    t_all = time.perf_counter()
    for _ in range(args.steps):
        a = time.perf_counter()
        opt.sync_lr(); model.draw_samples(n, step.samples)
        b = time.perf_counter()
        step._replay_device()
        c = time.perf_counter()
        td += b - a; tr += c - b
    host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    print(f"per step: draws+upload {td / args.steps * 1e3:.2f} ms, graph.replay() call {tr / args.steps * 1e3:.2f} ms, "
          f"host total {host / args.steps * 1e3:.2f} ms, wall {wall / args.steps * 1e3:.2f} ms")
    # GPU-side duration of one replay, alone (events around it; nothing else queued)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        model.draw_samples(n, step.samples); torch.cuda.synchronize()
        e0.record(); step._replay_device(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("GPU time of one isolated replay (ms):", [round(t, 2) for t in ts])


if __name__ == "__main__":
    main()
