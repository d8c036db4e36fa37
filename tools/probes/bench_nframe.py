"""Inference throughput of the n_frame model (model/test_DCNet_model.py semantics: centre frame against every other frame of its clip) at
BASELINE.json configs[3]'s geometry: clips of T = 16 frames of 608 x 608, 4 clips per batch, forward only, eval mode.
    python tools/bench_nframe.py [--size 608 --frames 16 --clips 4 --precision fp32|bf16s --iters 5]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dcnet_amd import ops
from dcnet_amd.utils.synth import synth_inputs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=608); ap.add_argument("--frames", type=int, default=16); ap.add_argument("--clips", type=int, default=4)
    ap.add_argument("--precision", default="fp32"); ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.set_precision(a.precision)
    torch.manual_seed(0)
    from model.test_DCNet_model import grounding_model as GM          # the drop-in of the reference's inference model
    m = GM(corpus=list(range(1000)), light=False, emb_size=512, coordmap=True, bert_model="bert-base-uncased", dataset="vid", img_size=a.size,
           config_path=os.path.join(ROOT, "model", "yolov3.cfg"), weights_path=None).to(dev).eval()
    image, word_id, word_mask = synth_inputs(a.clips * a.frames, a.size, n_queries=a.clips, seed=3)
    image, word_id, word_mask = image.to(dev), word_id.to(dev), word_mask.to(dev)
    with torch.no_grad():
        for _ in range(2):
            out = m(image, word_id, word_mask, a.frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = m(image, word_id, word_mask, a.frames)
        host = (time.perf_counter() - t0) / a.iters          # the Python call returns when everything is queued
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
    print(json.dumps({"workload": f"n_frame={a.frames} {a.size}x{a.size} {a.clips} clips/batch, eval forward", "precision": a.precision,
                      "ms_per_batch": round(dt * 1e3, 2), "host_queue_ms": round(host * 1e3, 2), "clips_per_s": round(a.clips / dt, 2), "frames_per_s": round(a.clips * a.frames / dt, 1),
                      "mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1), "finite": bool(all(torch.isfinite(o).all() for o in out[0]))}))


if __name__ == "__main__":
    main()
