"""Top-k candidate cache + temporal post-processing: the oracle, the product's HIP kernels
(csrc/post.hip) and the stock-torch restatement (tests/post_torch.py, the second opinion) against the fixtures captured from
the reference's get_topk_pred_bbox / post_processing (oracle/make_post_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from dcnet_amd import postprocess as PP
import post_torch as PT
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


def _product_vs_fixture(size, dev, tmp_path, topk_fn, fusion_fn, cache_files=True):
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
    boxes, score, feat, cells = topk_fn(outbox, corr, size, topk, ratio, dw, dh, hw)
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
    best, fused = fusion_fn(feat[items], rf, rs, valid)
    for n, it in enumerate(items):
        assert int(best[n]) == int(g[f"fuse_idx{it}"])
        assert np.allclose(fused[n].cpu().numpy(), g[f"fuse_scores{it}"], atol=1e-5)
    if not cache_files:
        return
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
def test_torch_second_opinion_matches_reference_fixture(size, tmp_path):
    """The stock-torch restatement the GPU tests compare the kernels with is itself pinned by the reference's fixture."""
    _product_vs_fixture(size, torch.device("cpu"), tmp_path, PT.topk_candidates_torch, PT.temporal_fusion_torch, cache_files=False)


def test_product_has_no_cpu_path():
    z = torch.zeros(1, 15, 8, 8)
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        PP.topk_candidates([z, z, z], [z, z, z], 256, 3, torch.ones(1), torch.zeros(1), torch.zeros(1), torch.zeros(1, 2, dtype=torch.long))
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        PP.temporal_fusion(torch.zeros(1, 3, 8), torch.zeros(1, 2, 3, 8), torch.zeros(1, 2, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [256, 416])
def test_product_matches_reference_fixture_gpu(size, tmp_path):
    _product_vs_fixture(size, torch.device("cuda:0"), tmp_path, PP.topk_candidates, PP.temporal_fusion)


@pytest.mark.gpu
def test_post_kernels_against_the_torch_forms():
    """csrc/post.hip against tests/post_torch.py on shapes the fixtures do not have: k up to 64, E not a multiple of 64, ties in
    the confidence (lowest flat index first, the reference's first exact match), NHWC-strided features, missing neighbours,
    a batch of windows; bitwise repeatable."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    for size, B, E, k in ((256, 3, 512, 5), (416, 2, 96, 64), (608, 2, 40, 17)):
        grids = [size // 32, size // 16, size // 8]
        outbox = [torch.randn(B, 15, gg, gg, generator=g).to(dev) for gg in grids]
        outbox[1][:, 4, 2, 3] = 9.0; outbox[1][:, 9, 1, 1] = 9.0; outbox[0][:, 14, 0, 5] = 9.0        # three-way tie at the top
        corr = [torch.randn(B, gg, gg, E, generator=g).to(dev).permute(0, 3, 1, 2) for gg in grids]
        ratio = (0.5 + torch.rand(B, generator=g)).to(dev); dw = (8 * torch.rand(B, generator=g)).to(dev); dh = (8 * torch.rand(B, generator=g)).to(dev)
        hw = torch.tensor([[size - 20, size - 8]] * B, device=dev)
        b1, s1, f1, c1 = PP.topk_candidates(outbox, corr, size, k, ratio, dw, dh, hw)
        b2, s2, f2, c2 = PT.topk_candidates_torch(outbox, corr, size, k, ratio, dw, dh, hw)
        assert torch.equal(s1, s2) and torch.all(s1[:, :-1] >= s1[:, 1:])
        assert c1[:, :3].tolist() == [[[0, 2, 0, 5], [1, 0, 2, 3], [1, 1, 1, 1]]] * B             # ties: flat-index order
        assert torch.equal(c1[:, 3:], c2[:, 3:]) and torch.equal(f1[:, 3:], f2[:, 3:])              # (torch.topk orders ties its own way)
        assert (b1[:, 3:] - b2[:, 3:]).abs().max() < 1e-3
        again = PP.topk_candidates(outbox, corr, size, k, ratio, dw, dh, hw)
        assert all(torch.equal(x, y) for x, y in zip((b1, s1, f1, c1), again))
        R = 5
        ref = torch.randn(B, R, k, E, generator=g).to(dev); cen = ref[:, R // 2].contiguous(); rs = torch.rand(B, R, k, generator=g).to(dev)
        valid = torch.ones(B, R, dtype=torch.bool, device=dev); valid[0, 0] = False; valid[-1, R - 1] = False
        for v in (None, valid):
            i1, u1 = PP.temporal_fusion(cen, ref, rs, v)
            i2, u2 = PT.temporal_fusion_torch(cen, ref, rs, v)
            assert (u1 - u2).abs().max() < 1e-5 and torch.equal(i1, i2)
            i3, u3 = PP.temporal_fusion(cen, ref, rs, v)
            assert torch.equal(u1, u3) and torch.equal(i1, i3)


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
