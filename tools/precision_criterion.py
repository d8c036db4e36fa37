"""SURVEY.md §8(c) box criterion for the reduced-precision modes, on weights that are not random.

The synthetic random-init network amplifies an operand rounding over ~75 layers, so `bf16` / `fp8` outputs cannot be compared
with fp32 outputs there (tests/test_model_gpu.py::test_reduced_precision_modes_end_to_end only bounds them).  Here the fp32
model is first trained with the product's own train step (RMSprop, the five losses) on a small fixed synthetic set whose images
carry the target — a textured rectangle at the ground-truth box over weak noise — until the confidence map is peaked; then the
SAME weights are evaluated in fp32, bf16-operand, bf16-storage ("bf16s") and fp8-operand mode and the decoded boxes compared:

    criterion (builder-defined): IoU(box_mode, box_fp32) >= 0.95 and the same arg-max (scale, anchor, cell).

Usage (GPU box):  python tools/precision_criterion.py [--size 256] [--images 16] [--steps 300] [--out gpurun_out/criterion.json]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_set(n, size, seed):
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs
    image, word_id, word_mask = synth_inputs(n, size, seed=seed)
    bbox = synth_boxes(n, size, seed=seed)
    r = np.random.RandomState(seed + 99)
    image = image * 0.3
    for i in range(n):
        x1, y1, x2, y2 = (int(v) for v in bbox[i])
        tex = torch.from_numpy(r.standard_normal((3, 1, 1)).astype(np.float32)) * 0.5 + 2.0
        stripes = ((torch.arange(y1, y2 + 1).view(-1, 1) + torch.arange(x1, x2 + 1).view(1, -1)) % 8 < 4).float() * 0.5 + 0.75
        image[i, :, y1:y2 + 1, x1:x2 + 1] += tex * stripes
    return image, word_id, word_mask, bbox


def argmax_cells(outbox):
    n = outbox[0].shape[0]
    conf = torch.cat([o.reshape(n, 3, 5, o.shape[-2], o.shape[-1])[:, :, 4].reshape(n, -1) for o in outbox], 1)
    return conf.argmax(1), conf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256); ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--steps", type=int, default=300); ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--out", type=str, default=os.path.join(ROOT, "gpurun_out", "criterion.json"))
    ap.add_argument("--regions", action="store_true",
                    help="localise the bf16-storage drift: evaluate with ONE region of the forward at a time in fp32 (rest bf16s) and with one "
                         "region at a time in bf16s (rest fp32) — ops.REGION_PRECISION")
    ap.add_argument("--backbone", action="store_true",
                    help="localise the drift INSIDE the bf16-storage backbone (eval): fp32 shortcut sums, fp32 for the layers before / from a "
                         "stage boundary — ops.B16_DIAG")
    ap.add_argument("--no-train-in-mode", action="store_true", help="skip the three trainings IN the reduced modes")
    args = ap.parse_args()
    from dcnet_amd import losses, ops, train as T
    from dcnet_amd.parallel import freeze_gradless
    from util import build_product, synth_sd
    dev = torch.device("cuda:0")
    size, n = args.size, args.images
    m = build_product(size, synth_sd(size), dev)
    freeze_gradless(m)
    opt = T.make_optimizer(m, args.lr)
    image, word_id, word_mask, bbox = (t.to(dev) for t in make_set(n, size, 5))
    random.seed(0)
    t0 = time.time(); hist = []
    for it in range(args.steps):
        T.adjust_learning_rate(opt, it, args.lr, args.steps, 0.9)
        loss, parts = T.train_step(m, opt, image, word_id, word_mask, bbox, size)
        if it % 25 == 0 or it == args.steps - 1:
            hist.append((it, float(loss)))
            print(f"step {it:4d} loss {float(loss):.4f}  " + " ".join(f"{k}={float(v):.4f}" for k, v in parts.items()), flush=True)
    train_s = time.time() - t0
    res = {}
    m.eval()
    try:
        for mode in ("fp32", "bf16", "bf16s", "fp8", "fp8s"):
            ops.set_precision(mode)
            with torch.no_grad():
                outbox = [o.float() for o in m(image, word_id, word_mask)[0]]
            boxes = losses.decode_boxes(outbox, size)
            cell, conf = argmax_cells(outbox)
            iou_gt = losses.bbox_iou(boxes, torch.clamp(bbox, min=0, max=size - 1))
            res[mode] = dict(boxes=boxes, cell=cell, conf=conf, iou_gt=iou_gt)
    finally:
        ops.set_precision("fp32")
    out = {"size": size, "images": n, "train_steps": args.steps, "train_seconds": train_s, "loss_history": hist,
           "criterion": "IoU(box_mode, box_fp32) >= 0.95 and same arg-max (scale, anchor, cell)", "modes": {}}
    f = res["fp32"]
    top2 = f["conf"].topk(2, 1).values
    out["fp32"] = {"acc_at_0.5_vs_gt": float((f["iou_gt"] > 0.5).float().mean()), "mean_iou_vs_gt": float(f["iou_gt"].mean()),
                   "conf_margin_top1_minus_top2_min": float((top2[:, 0] - top2[:, 1]).min()),
                   "conf_margin_top1_minus_top2_median": float((top2[:, 0] - top2[:, 1]).median())}
    for mode in ("bf16", "bf16s", "fp8", "fp8s"):
        r = res[mode]
        iou = losses.bbox_iou(r["boxes"], f["boxes"])
        same = (r["cell"] == f["cell"])
        ok = (iou >= 0.95) & same
        out["modes"][mode] = {"acc_at_0.5_vs_gt": float((r["iou_gt"] > 0.5).float().mean()), "mean_iou_vs_gt": float(r["iou_gt"].mean()),
                              "iou_vs_fp32_min": float(iou.min()), "iou_vs_fp32_mean": float(iou.mean()),
                              "same_argmax_cell_frac": float(same.float().mean()), "criterion_met_frac": float(ok.float().mean()),
                              "max_abs_conf_diff": float((r["conf"] - f["conf"]).abs().max())}
    if args.regions:
        names = ("language", "backbone", "mapping", "corr", "fusion", "out", "tail")
        table = {}
        try:
            cases = {"all_bf16s": {"default": "bf16s"}, "all_fp32": {"default": "fp32"}}
            for r_ in names:
                cases[f"fp32_only_{r_}"] = {"default": "bf16s", r_: "fp32"}
                cases[f"bf16s_only_{r_}"] = {"default": "fp32", r_: "bf16s"}
            cases["fp32_fusion+out"] = {"default": "bf16s", "fusion": "fp32", "out": "fp32"}
            cases["fp32_head(all_but_backbone)"] = {"default": "fp32", "backbone": "bf16s"}
            for cname, rp in cases.items():
                ops.REGION_PRECISION = rp
                with torch.no_grad():
                    outbox = [o.float() for o in m(image, word_id, word_mask)[0]]
                boxes = losses.decode_boxes(outbox, size)
                cell, conf = argmax_cells(outbox)
                iou = losses.bbox_iou(boxes, f["boxes"])
                same = (cell == f["cell"])
                table[cname] = {"met": int(((iou >= 0.95) & same).sum()), "iou_min": round(float(iou.min()), 4), "iou_mean": round(float(iou.mean()), 4),
                                "same_cell": int(same.sum())}
                print(f"regions {cname:34s} {table[cname]}", flush=True)
        finally:
            ops.REGION_PRECISION = None
            ops.set_precision("fp32")
        out["regions"] = table
    if args.backbone:
        table = {}
        cases = {"bf16s": None, "res32": {"res32": True}}
        for k_ in (5, 12, 37, 62, 75):          # first conv slot of the 104^2 / 52^2 / 26^2 / 13^2 stages, end of the trunk
            cases[f"fp32_before_{k_}"] = {"layers": (lambda s_, k=k_: s_ >= k)}
            cases[f"fp32_from_{k_}"] = {"layers": (lambda s_, k=k_: s_ < k)}
            cases[f"res32+fp32_before_{k_}"] = {"layers": (lambda s_, k=k_: s_ >= k), "res32": True}
        for set_ in ((1,), (1, 2), (1, 2, 3), (1, 2, 3, 4), (2, 3, 4), (3, 4), (4,), (1, 4), (1, 2, 3, 4, 5), (5, 6, 7, 8, 9, 10, 11)):
            cases["fp32_slots_" + "_".join(map(str, set_))] = {"layers": (lambda s_, q=set_: s_ not in q)}
        for set_ in ((1, 2, 3, 4), (1,), (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11)):
            cases["fp32tensors_bf16ops_slots_" + "_".join(map(str, set_))] = {"layers": (lambda s_, q=set_: s_ not in q), "arith": "bf16"}
        try:
            ops.set_precision("bf16s")
            for cname, dg in cases.items():
                ops.B16_DIAG = None if dg is None else dict(dg)
                with torch.no_grad():
                    outbox = [o.float() for o in m(image, word_id, word_mask)[0]]
                boxes = losses.decode_boxes(outbox, size)
                cell, conf = argmax_cells(outbox)
                iou = losses.bbox_iou(boxes, f["boxes"])
                same = (cell == f["cell"])
                table[cname] = {"met": int(((iou >= 0.95) & same).sum()), "iou_min": round(float(iou.min()), 4), "iou_mean": round(float(iou.mean()), 4),
                                "same_cell": int(same.sum())}
                print(f"backbone {cname:34s} {table[cname]}", flush=True)
        finally:
            ops.B16_DIAG = None
            ops.set_precision("fp32")
        out["backbone"] = table
    # the other question of configs[2] / configs[4]: does TRAINING in the mode reach the same accuracy?  Same initial weights,
    # same data and schedule, every step in the mode; evaluated in the mode against the ground truth.
    out["trained_in_mode"] = {}
    for mode in (() if args.no_train_in_mode else ("bf16", "bf16s", "fp8", "fp8s")):
        try:
            ops.set_precision(mode)
            m2 = build_product(size, synth_sd(size), dev)
            freeze_gradless(m2)
            opt2 = T.make_optimizer(m2, args.lr)
            random.seed(0)
            for it in range(args.steps):
                T.adjust_learning_rate(opt2, it, args.lr, args.steps, 0.9)
                loss, parts = T.train_step(m2, opt2, image, word_id, word_mask, bbox, size)
            acc, miou, _ = T.evaluate(m2, image, word_id, word_mask, bbox, size)
            out["trained_in_mode"][mode] = {"final_loss": float(loss), "yolo": float(parts["yolo"]), "acc_at_0.5_vs_gt": float(acc),
                                            "mean_iou_vs_gt": float(miou)}
        finally:
            ops.set_precision("fp32")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
