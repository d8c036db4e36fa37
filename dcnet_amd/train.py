"""Training / evaluation harness around the hot path — the pieces of train_DCNet.py that a caller needs
to drive ``grounding_model`` without the reference's data pipeline:

  * ``make_optimizer``      RMSprop with the reference's two parameter groups (train_DCNet.py:519-534)
  * ``adjust_learning_rate`` polynomial decay (train_DCNet.py:241-253)
  * ``train_step``          forward + five losses + backward + step, no host sync inside
  * ``evaluate``            eval forward + box decode + Acc@0.5 / mean IoU (train_DCNet.py:764-816)
  * ``save_checkpoint`` / ``load_checkpoint`` / ``load_pretrain``   the reference's ``.pth.tar`` dict
    (train_DCNet.py:255-263, 485-514) including the ``module.`` key prefix left by DDP wrappers

``python -m dcnet_amd.train --steps 20`` runs a short synthetic-data training loop on one GPU.
"""
from __future__ import annotations

import argparse
import os
import random
import shutil
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import losses, ops


def make_optimizer(model, lr: float = 1e-4, optimizer: str = "RMSprop"):
    """Two groups like the reference: everything except the backbone at ``lr``, the Darknet backbone at
    ``lr / 10``; weight decay 5e-4 (train_DCNet.py:519-534).  ``model`` may be DDP-wrapped.

    Every parameter is listed, in ``model.parameters()`` order, whether or not it is trainable — the reference's
    groups hold [93, 222] tensors including the dead YOLO heads and ``feature_map`` — so that the ``optimizer`` entry
    of a ``.pth.tar`` checkpoint moves between the reference and this harness in both directions even after
    ``parallel.freeze_gradless`` has run.  Parameters without a gradient are skipped by the step, as in torch."""
    core = model.module if hasattr(model, "module") else model
    visu = list(core.visumodel.parameters())
    ids = {id(p) for p in visu}
    rest = [p for p in core.parameters() if id(p) not in ids]
    if optimizer.lower() == "adam":
        return torch.optim.Adam(list(core.parameters()), lr=lr, weight_decay=0.0005)
    if optimizer.lower() == "sgd":
        return torch.optim.SGD(list(core.parameters()), lr=lr, momentum=0.99)
    from .optim import RMSprop          # torch.optim.RMSprop's update as one fused HIP pass (same state_dict layout)
    return RMSprop([{"params": rest}, {"params": visu, "lr": lr / 10.}], lr=lr, weight_decay=0.0005)


def lr_poly(base_lr: float, it: int, max_iter: int, power: float) -> float:
    return base_lr * ((1 - float(it) / max_iter) ** power)                      # train_DCNet.py:241-242


def adjust_learning_rate(optimizer, i_iter: int, base_lr: float, nb_epoch: int, power: float = 0.9) -> float:
    """train_DCNet.py:244-253: group 0 at the polynomial rate, group 1 (backbone) at a tenth of it."""
    lr = lr_poly(base_lr, i_iter, nb_epoch, power) if power != 0. else base_lr
    optimizer.param_groups[0]["lr"] = lr
    if len(optimizer.param_groups) > 1:
        optimizer.param_groups[1]["lr"] = lr / 10
    return lr


def train_step(model, optimizer, image, word_id, word_mask, bbox, size: int):
    """One optimisation step (train_DCNet.py:580-646).  Returns (loss, dict of the five parts) as device
    tensors — reading them is the caller's (only) synchronisation point."""
    model.train()
    out = model(image, word_id, word_mask)
    loss, parts = losses.total_loss(out, bbox, size)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    return loss.detach(), {k: v.detach() for k, v in parts.items()}


@torch.no_grad()
def evaluate(model, image, word_id, word_mask, bbox, size: int, n_frame: Optional[int] = None):
    """Eval forward, decode the arg-max box, IoU against ``bbox`` (xyxy pixels).  Returns
    (acc@0.5, mean IoU, boxes) as tensors (train_DCNet.py:764-816 / test_DCNet.py:373-470)."""
    model.eval()
    if n_frame is None:
        outbox = model(image, word_id, word_mask)[0]
    else:
        outbox = model(image, word_id, word_mask, n_frame)[0]
    boxes = losses.decode_boxes(list(outbox), size)
    iou = losses.bbox_iou(boxes, torch.clamp(bbox, min=0, max=size - 1))
    ops.check_bilstm(image.device)              # (evaluation results are read by the host anyway: one device synchronisation)
    return (iou > 0.5).float().mean(), iou.mean(), boxes


# ---- checkpoints ------------------------------------------------------------------------------------
def _strip_module(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def save_checkpoint(state: dict, is_best: bool, filename: str, directory: str = "./saved_models") -> str:
    """Writes ``<dir>/<filename>_checkpoint.pth.tar`` (+ ``_model_best`` copy), train_DCNet.py:255-263.
    ``state`` = {'epoch', 'state_dict', 'best_loss', 'optimizer'}."""
    for t_ in state.get("state_dict", {}).values():
        if torch.is_tensor(t_) and t_.is_cuda:
            ops.check_bilstm(t_.device)         # (host-synchronising anyway: nothing is written after a timed-out BiLSTM hand-off)
            break
    os.makedirs(directory, exist_ok=True)
    ckpt = os.path.join(directory, f"{filename}_checkpoint.pth.tar")
    torch.save(state, ckpt)
    if is_best:
        shutil.copyfile(ckpt, os.path.join(directory, f"{filename}_model_best.pth.tar"))
    return ckpt


def load_checkpoint(model, path: str, optimizer=None, map_location="cpu") -> Tuple[int, float]:
    """``--resume`` (train_DCNet.py:500-514): strict load of model (+ optimizer).  Accepts checkpoints saved
    from a DDP/DataParallel wrapper (``module.`` prefix) into a bare model and vice versa."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    core = model.module if hasattr(model, "module") else model
    core.load_state_dict(_strip_module(ck["state_dict"]), strict=True)
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    return int(ck.get("epoch", 0)), float(ck.get("best_loss", float("inf")))


def load_pretrain(model, path: str, map_location="cpu") -> int:
    """``--pretrain`` (train_DCNet.py:485-499): load the intersection of keys with matching shapes.
    Returns the number of tensors taken."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    src = _strip_module(ck["state_dict"] if "state_dict" in ck else ck)
    core = model.module if hasattr(model, "module") else model
    own = core.state_dict()
    take = {k: v for k, v in src.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    own.update(take)
    core.load_state_dict(own, strict=True)
    return len(take)


def main(argv: Optional[Iterable[str]] = None) -> None:
    ap = argparse.ArgumentParser(description="short synthetic-data training run of the HIP-backed DCNet")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--lr", type=float, default=1e-4)
    args = ap.parse_args(argv)
    from .model import grounding_model
    from .parallel import freeze_gradless
    from .utils.synth import synth_boxes, synth_inputs
    dev = torch.device("cuda:0")
    torch.manual_seed(0); random.seed(0)
    model = grounding_model(corpus=list(range(1000)), emb_size=512, img_size=args.size, config_path="", weights_path=None).to(dev)
    freeze_gradless(model)
    opt = make_optimizer(model, args.lr)
    n = args.clips * args.frames
    image, word_id, word_mask = (t.to(dev) for t in synth_inputs(n, args.size, seed=1))
    bbox = synth_boxes(n, args.size, seed=1).to(dev)
    for it in range(args.steps):
        adjust_learning_rate(opt, it, args.lr, args.steps, 0.9)
        loss, parts = train_step(model, opt, image, word_id, word_mask, bbox, args.size)
        if it % 5 == 0 or it == args.steps - 1:
            print(f"step {it:3d} loss {float(loss):9.4f}  " + " ".join(f"{k} {float(v):.4f}" for k, v in parts.items()))
    acc, miou, _ = evaluate(model, image, word_id, word_mask, bbox, args.size)
    print(f"Acc@0.5 {float(acc):.3f}  mIoU {float(miou):.3f} (on the training clips)")


if __name__ == "__main__":
    main()
