"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import json
import os
import random

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def num_pos(size):
    return sum((size // 32 * 2 ** i) ** 2 for i in range(3))


def ref_shapes(size=256):
    """The reference's 597 state_dict keys/shapes (captured from the real reference at 256x256);
    only loc_text_embedding.0.weight depends on the image size (P = num_pos)."""
    with open(os.path.join(GOLD, "state_dict_keys_256.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    shapes["loc_text_embedding.0.weight"] = (512, num_pos(size))
    return shapes


def synth_sd(size=256, seed=0):
    from dcnet_amd.utils.synth import apply_bn_calibration, synth_state_dict
    return apply_bn_calibration(synth_state_dict(ref_shapes(size), seed), os.path.join(GOLD, "bn_calib.npz"))


def build_product(size, sd, dev, test_model=False):
    if test_model:
        from model.test_DCNet_model import grounding_model
    else:
        from model.DCNet_model import grounding_model
    m = grounding_model(corpus=list(range(1000)), light=False, emb_size=512, coordmap=True,
                        bert_model="bert-base-uncased", dataset="vid", img_size=size,
                        config_path=os.path.join(ROOT, "model", "yolov3.cfg"), weights_path=None)
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0           # parity runs use p = 0 on both sides
    return m.to(dev)


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())
