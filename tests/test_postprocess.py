"""Top-k candidate cache + temporal post-processing: oracle and batched product
against the fixtures captured from the reference's get_topk_pred_bbox / post_processing
(oracle/make_post_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from dcnet_amd import postprocess as PP
from dcnet_amd.utils.synth import synth_head_outputs
from oracle import post_oracle as PO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(size):
    g = np.load(os.path.join(GOLD, f"post_S{size}.npz"))
    return g, int(g["E"]), int(g["topk"]), int(g["n_items"]), int(g["num_frame_k"])


@pytest.mark.parametrize("size", [256, 416])
def test_oracle_matches_reference_fixture(size):
    g, E, topk, n_items, nk = _load(size)
    entries = []
    for it in range(n_items):
        pred, feat = synth_head_outputs(size, E, 1000 * size + it)
        ratio, dw, dh, H, W = g[f"meta{it}"]
        assert PO.letterbox_frame(size, ratio, dw, dh) == (int(H), int(W))
        b, s, f, cells = PO.topk_candidates(pred, feat, size, topk, float(ratio), float(dw), float(dh))
        assert np.array_equal(b.numpy(), g[f"boxes{it}"])
        assert np.array_equal(np.array(s, dtype=np.float32), g[f"scores{it}"])
        assert np.array_equal(np.array(cells), g[f"cells{it}"])
        entries.append((b, s, f))
    c = nk // 2
    for it in range(n_items):
        if f"fuse_idx{it}" not in g:
            continue
        inv = list(g[f"fuse_invalid{it}"])
        rf, rs = [], []
        for o, frm in zip(range(-c, c + 1), range(nk)):
            src = entries[it] if frm in inv else entries[it + o]
            rf.append(src[2]); rs.append(torch.tensor(src[1], dtype=torch.float))
        idx, fused = PO.temporal_fusion(entries[it][2], rf, rs, inv)
        assert idx == int(g[f"fuse_idx{it}"])
        assert np.allclose(fused.numpy(), g[f"fuse_scores{it}"], atol=1e-6)
        assert np.array_equal(entries[it][0][idx].numpy(), g[f"fuse_box{it}"])


def _product_vs_fixture(size, dev, tmp_path):
    g, E, topk, n_items, nk = _load(size)
    preds, feats = zip(*[synth_head_outputs(size, E, 1000 * size + it) for it in range(n_items)])
    outbox = [torch.cat([p[s] for p in preds]).to(dev) for s in range(3)]                 # all clips in one batch
    # features arrive as NHWC-strided views from the model: exercise that
    corr = [torch.cat([f[s] for f in feats]).to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for s in range(3)]
    meta = np.stack([g[f"meta{it}"] for it in range(n_items)])
    ratio, dw, dh = (torch.tensor(meta[:, i], dtype=torch.float32, device=dev) for i in range(3))
    hw = torch.tensor(meta[:, 3:5], dtype=torch.int64, device=dev)
    for it in range(n_items):
        assert PP.letterbox_frame(size, meta[it, 0], meta[it, 1], meta[it, 2]) == (int(meta[it, 3]), int(meta[it, 4]))
    boxes, score, feat, cells = PP.topk_candidates(outbox, corr, size, topk, ratio, dw, dh, hw)
    for it in range(n_items):
        assert np.array_equal(cells[it].cpu().numpy(), g[f"cells{it}"])                 # integer work: exact
        assert np.array_equal(score[it].cpu().numpy(), g[f"scores{it}"])
        assert np.abs(boxes[it].cpu().numpy() - g[f"boxes{it}"][:, 0]).max() < 1e-3      # pixels
        ref_feat = torch.stack([feats[it][int(s)][0, :, int(j), int(i)] for s, _, j, i in g[f"cells{it}"]])
        assert torch.equal(feat[it].cpu(), ref_feat)
    # batched fusion: all windows at once
    c = nk // 2
    items = [it for it in range(n_items) if f"fuse_idx{it}" in g]
    rf = torch.zeros(len(items), nk, topk, E, device=dev); rs = torch.zeros(len(items), nk, topk, device=dev)
    valid = torch.ones(len(items), nk, dtype=torch.bool, device=dev)
    for n, it in enumerate(items):
        inv = list(g[f"fuse_invalid{it}"])
        for o, frm in zip(range(-c, c + 1), range(nk)):
            j = it if frm in inv else it + o
            rf[n, frm] = feat[j]; rs[n, frm] = score[j]; valid[n, frm] = frm not in inv
    best, fused = PP.temporal_fusion(feat[items], rf, rs, valid)
    for n, it in enumerate(items):
        assert int(best[n]) == int(g[f"fuse_idx{it}"])
        assert np.allclose(fused[n].cpu().numpy(), g[f"fuse_scores{it}"], atol=1e-5)
    # the reference's cache files: write entries (item 1 missing, as in the fixture), fuse from disk
    names = [f"/data/vid{size}/{i:06d}.JPEG" for i in range(n_items)]
    for it in items:
        PP.save_cache_entry(PP.cache_file(str(tmp_path), names[it], it), boxes[it], score[it], feat[it])
    b0, s0, f0 = PP.load_cache_entry(PP.cache_file(str(tmp_path), names[items[0]], items[0]))
    assert b0.shape == (topk, 1, 4) and s0.shape == (topk,) and f0.shape == (topk, 1, E)
    for it in items:
        ids = [names[min(max(it + o, 0), n_items - 1)] for o in range(-c, c + 1)][:nk]
        box, idx, _ = PP.fuse_from_cache(str(tmp_path), ids, it, nk, device=dev)
        assert idx == int(g[f"fuse_idx{it}"])
        assert np.abs(box.numpy() - g[f"fuse_box{it}"]).max() < 1e-3


@pytest.mark.parametrize("size", [256, 416])
def test_product_matches_reference_fixture_cpu(size, tmp_path):
    _product_vs_fixture(size, torch.device("cpu"), tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [256, 416])
def test_product_matches_reference_fixture_gpu(size, tmp_path):
    _product_vs_fixture(size, torch.device("cuda:0"), tmp_path)


@pytest.mark.gpu
def test_topk_candidates_on_model_outputs():
    """End to end on the HIP model: n_frame eval forward -> top-k candidates; the first candidate must be the
    box the plain evaluation decode picks, and features must be the corr_feat columns of the winning cells."""
    from dcnet_amd import losses
    from dcnet_amd.model import grounding_model
    from dcnet_amd.utils.synth import synth_inputs
    dev = torch.device("cuda:0")
    size, T, B = 256, 3, 2
    torch.manual_seed(0)
    m = grounding_model(corpus=list(range(1000)), emb_size=512, img_size=size, config_path="", weights_path=None).to(dev).eval()
    image, word_id, word_mask = synth_inputs(B * T, size, n_queries=B, seed=5)
    with torch.no_grad():
        outbox, sim, loc, corr, only_obj = m(image.to(dev), word_id.to(dev), word_mask.to(dev), T)
    one = torch.ones(B, device=dev); zero = torch.zeros(B, device=dev)
    hw = torch.full((B, 2), size, device=dev)
    boxes, score, feat, cells = PP.topk_candidates(list(outbox), list(corr), size, 4, one, zero, zero, hw)
    top1 = losses.decode_boxes(list(outbox), size)
    assert torch.allclose(boxes[:, 0], torch.minimum(top1.clamp(min=0), torch.tensor(float(size), device=dev)), atol=1e-3)
    assert torch.all(score[:, :-1] >= score[:, 1:])
    for b in range(B):
        for k in range(4):
            s_, _, j, i = (int(v) for v in cells[b, k])
            assert torch.equal(feat[b, k], corr[s_][b, :, j, i])
