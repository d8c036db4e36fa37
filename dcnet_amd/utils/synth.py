"""Version-stable synthetic weights and inputs (no dataset / checkpoint exists offline).

Values come from ``numpy.random.RandomState`` (the frozen legacy MT19937 stream),
seeded per state_dict key from crc32(key), so a tensor's values depend only on
(base seed, key, shape) — never on dict order or on the torch version.  Tests,
bench.py and oracle/make_goldens.py all build their weights with this, which is
how a golden fixture can be re-derived on the GPU box without shipping 323 MB.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch


def _rs(seed: int, key: str) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """One parameter/buffer of the DCNet state_dict, scaled so that activations stay
    O(1) through 75 conv layers (He fan-in for convs, 1/sqrt(fan_in) for linears)."""
    r = _rs(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    shape = tuple(shape)
    n = int(np.prod(shape)) if shape else 1
    is_bn = (".bn." in key or "batch_norm" in key or key.startswith(("mapping_lang.1.", "mapping_lang.5.",
             "loc_embedding.1.", "loc_text_embedding.1.")))
    if leaf == "running_mean":
        a = r.normal(0.0, 0.1, n)
    elif leaf == "running_var":
        a = r.uniform(0.8, 1.2, n)
    elif is_bn and leaf == "weight":
        a = r.uniform(0.8, 1.2, n)
    elif is_bn and leaf == "bias":
        a = r.uniform(-0.1, 0.1, n)
    elif ".rnn." in key:
        h = 512
        a = r.uniform(-1.0 / np.sqrt(h), 1.0 / np.sqrt(h), n)
    elif "embedding.weight" in key and len(shape) == 2 and key.startswith("textmodel"):
        a = r.normal(0.0, 1.0, n)
    elif len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        a = r.normal(0.0, np.sqrt(2.0 / fan_in), n)
    elif len(shape) == 3:
        a = r.normal(0.0, 1.0 / np.sqrt(shape[1] * shape[2]), n)
    elif len(shape) == 2:
        a = r.normal(0.0, 1.0 / np.sqrt(shape[1]), n)
    else:
        a = r.uniform(-0.1, 0.1, n)
    return torch.from_numpy(a.astype(np.float32)).reshape(shape)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, synth_tensor(k, s, seed)) for k, s in shapes.items())


def synth_inputs(n_images: int, size: int, n_queries: int | None = None, query_len: int = 20,
                 vocab: int = 1000, n_words: int | None = None, seed: int = 0):
    """image (N,3,S,S) fp32 ~ N(0,1); word_id (Q,L) int64 with every id non-zero
    (F4: the reference tokenizer pads with non-zero <eos>/<pad> ids, so L is always
    20).  ``n_words`` < L emulates a short query: ids[n_words]=eos, rest=pad."""
    r = np.random.RandomState(1000003 * seed + 17)
    q = n_images if n_queries is None else n_queries
    image = torch.from_numpy(r.standard_normal((n_images, 3, size, size)).astype(np.float32))
    ids = r.randint(3, vocab, size=(q, query_len)).astype(np.int64)
    if n_words is not None and n_words < query_len:
        ids[:, n_words] = 2          # <eos>
        ids[:, n_words + 1:] = 1     # <pad>
    word_id = torch.from_numpy(ids)
    word_mask = torch.zeros_like(word_id)
    return image, word_id, word_mask


def synth_boxes(n: int, size: int, seed: int = 0) -> torch.Tensor:
    """GT boxes (x1,y1,x2,y2) uniform in [0,S) with min side 16 px."""
    r = np.random.RandomState(424243 * seed + 5)
    x1 = r.uniform(0, size - 17, n); y1 = r.uniform(0, size - 17, n)
    w = r.uniform(16, size - 1 - x1); h = r.uniform(16, size - 1 - y1)
    return torch.from_numpy(np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32))


def apply_bn_calibration(sd, npz_path: str):
    """Overwrite the BatchNorm running stats of ``sd`` with the calibrated ones in
    tests/golden/bn_calib.npz (recorded from one train-mode pass of the reference on
    the synthetic weights, oracle/make_goldens.py step 0b).  Without it the eval-mode
    residual trunk of a randomly initialised Darknet-53 grows as 2^23."""
    with np.load(npz_path) as z:
        for k in z.files:
            if k in sd and tuple(sd[k].shape) == z[k].shape:
                sd[k] = torch.from_numpy(z[k].copy())
    return sd


def synth_head_outputs(size: int, emb: int, seed: int):
    """Seeded stand-ins for one clip's modulated head outputs (1,15,g,g) x3 and unit-norm correspondence
    features (1,emb,g,g) x3, for the top-k cache / post-processing tests (numpy's legacy RandomState
    stream is version-frozen, so fixtures need not store them)."""
    rs = np.random.RandomState(seed)
    grids = [size // 32, size // 16, size // 8]
    pred = [torch.from_numpy(rs.standard_normal((1, 15, g, g)).astype(np.float32)) for g in grids]
    feat = [torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((1, emb, g, g)).astype(np.float32)), dim=1)
            for g in grids]
    return pred, feat
