"""Pin oracle/post_oracle.py to the REAL reference's top-k cache + temporal post-processing and write
tests/golden/post_S*.npz.  Build container only (needs /root/reference):

    python -m oracle.make_post_goldens

The reference functions are driven directly:
  * ``test_DCNet.get_topk_pred_bbox`` (test_DCNet.py:662-705) on seeded synthetic head outputs;
  * ``post_processing.post_processing`` (post_processing.py:193-352) over a fake loader and a temporary
    cache directory written in ``save_cache``'s format; the box it selects is observed by wrapping the
    ``bbox_iou`` name it calls at :320.
Stubs: the import shells of oracle/make_goldens.py plus ``cv2.resize`` returning a zero image of the
requested size (only its ``.shape`` is consumed, for the clamp at test_DCNet.py:700-701).
Fixtures hold the reference's outputs; inputs are regenerated from seeds (dcnet_amd.utils.synth.synth_head_outputs).
"""
from __future__ import annotations

import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

from . import make_goldens as MG

GOLD = MG.GOLD


def main():
    MG._import_reference()
    import cv2
    cv2.INTER_CUBIC = 2
    cv2.resize = lambda img, shape, interpolation=None: np.zeros((shape[1], shape[0], 3), dtype=np.float32)
    sys.path.insert(0, MG.ROOT)
    import test_DCNet as T                    # noqa: E402  (the reference's script, imported as a module)
    import post_processing as PP              # noqa: E402
    from oracle import post_oracle as PO
    from dcnet_amd.utils.synth import synth_head_outputs

    anchors = [float(x) for x in '10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326'.split(',')]
    anchors_full = [(anchors[i], anchors[i + 1]) for i in range(0, len(anchors), 2)][::-1]
    assert anchors_full == PO.ANCHORS_FULL

    for size, E, topk, n_items, nk in ((256, 32, 5, 6, 5), (416, 16, 3, 4, 3)):
        T.args = SimpleNamespace(size=size, anchor_imsize=416)
        T.anchors_full = anchors_full
        rng = np.random.RandomState(size)
        out = {"size": size, "E": E, "topk": topk, "n_items": n_items, "num_frame_k": nk}
        entries = []
        for it in range(n_items):
            pred, feat = synth_head_outputs(size, E, 1000 * size + it)
            ratio = float(rng.uniform(0.3, 0.9)); dwv = float(rng.choice([0.0, 12.5, 31.0])); dhv = 0.0 if dwv else float(rng.choice([0.0, 20.5]))
            dw, dh = torch.tensor([dwv]), torch.tensor([dhv])
            H, W = PO.letterbox_frame(size, ratio, dwv, dhv)
            img_np = torch.zeros(1, 3, H, W)
            pa = [p.view(1, 3, 5, p.size(2), p.size(3)) for p in pred]
            conf_list = [p[:, :, 4, :, :].contiguous().view(1, -1) for p in pa]
            mc, ml = torch.topk(torch.cat(conf_list, dim=1), k=topk, dim=1)
            rb, rs, rf, rc = [], [], [], []
            for ii in range(topk):
                bb, sc_, bs_, bn_, gj_, gi_ = T.get_topk_pred_bbox(conf_list, None, None, mc[:, ii], ml[:, ii], pa, ratio, dw, dh, img_np)
                rb.append(bb); rs.append(float(sc_)); rf.append(feat[bs_][:, :, gj_, gi_]); rc.append((bs_, bn_, gj_, gi_))
            ob, os_, of, oc = PO.topk_candidates(pred, feat, size, topk, ratio, dwv, dhv)
            assert torch.equal(torch.stack(rb), ob), (torch.stack(rb), ob)
            assert rs == os_ and rc == oc and torch.equal(torch.stack(rf), of)
            entries.append((torch.stack(rb), rs, torch.stack(rf)))
            out[f"meta{it}"] = np.array([ratio, dwv, dhv, H, W], dtype=np.float64)
            out[f"boxes{it}"] = torch.stack(rb).numpy(); out[f"scores{it}"] = np.array(rs, dtype=np.float32)
            out[f"cells{it}"] = np.array(rc, dtype=np.int64)
        print(f"S={size}: get_topk_pred_bbox == oracle on {n_items} clips x top-{topk}")

        # ---- temporal fusion through the reference's post_processing() -----------------------------
        names = [f"/data/vid{size}/{i:06d}.JPEG" for i in range(n_items)]
        picked = []
        with tempfile.TemporaryDirectory() as tmp:
            for it, (b, s, f) in enumerate(entries):
                if it == 1:
                    continue                                                    # a missing neighbour inside the window
                p = os.path.join(tmp, f"vid{size}", names[it].split("/")[-1].split(".JPEG")[0] + f"_{it}.pth")
                os.makedirs(os.path.dirname(p), exist_ok=True)
                torch.save({"pred_bbox_topk": b, "pred_score_topk": s, "visu_feat": f}, p)
            c = nk // 2
            items = [it for it in range(n_items) if it != 1]
            def loader():
                # batch_idx enumerates the loader, so yield a placeholder for every index and let the
                # windows of the missing item be skipped by the consumer below
                for it in range(n_items):
                    ids = [(names[min(max(it + o, 0), n_items - 1)],) for o in range(-c, c + 1)][:nk]
                    yield (torch.zeros(1, nk, 3, size, size), None, None, torch.tensor([[[10., 10., 100., 100.]] * nk]),
                           torch.full((1, nk), 0.5), torch.zeros(1, nk), torch.zeros(1, nk), ids, ["q"])
            class _L:
                def __len__(self): return n_items
            # batch_idx must keep the dataset numbering although item 1 is skipped: give the module its
            # own ``enumerate`` for this call
            PP.enumerate = lambda ld: ((i, x) for i, x in enumerate(loader()) if i != 1)
            PP.args = SimpleNamespace(num_frame_k=nk, cache_dir=tmp, size=size, save_plot=False, print_freq=10 ** 9, savename="g")
            PP.bbox_iou = lambda pb, tb, x1y1x2y2=True: (picked.append(pb.clone()), torch.zeros(1))[1]
            PP.post_processing(_L(), False, topk)
            del PP.enumerate
            assert len(picked) == len(items)
            for n, it in enumerate(items):
                ids = [names[min(max(it + o, 0), n_items - 1)] for o in range(-c, c + 1)][:nk]
                rf_, rs_, inv = [], [], []
                for o, frm in zip(range(-c, c + 1), range(nk)):
                    j = it + o
                    p = os.path.join(tmp, f"vid{size}", ids[frm].split("/")[-1].split(".JPEG")[0] + f"_{j}.pth")
                    src = entries[j] if os.path.exists(p) else entries[it]
                    if not os.path.exists(p):
                        inv.append(frm)
                    rf_.append(src[2]); rs_.append(torch.tensor(src[1], dtype=torch.float))
                idx, fused = PO.temporal_fusion(entries[it][2], rf_, rs_, inv)
                assert torch.equal(entries[it][0][idx], picked[n]), (it, idx, picked[n])
                out[f"fuse_idx{it}"] = np.int64(idx); out[f"fuse_box{it}"] = picked[n].numpy()
                out[f"fuse_scores{it}"] = fused.numpy(); out[f"fuse_invalid{it}"] = np.array(inv, dtype=np.int64)
        print(f"S={size}: post_processing() == oracle on {len(items)} windows (missing-neighbour windows included)")
        np.savez_compressed(os.path.join(GOLD, f"post_S{size}.npz"), **out)


if __name__ == "__main__":
    main()
