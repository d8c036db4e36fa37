"""Pin the oracle to the REAL reference and write tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference, never the GPU box):

    python -m oracle.make_goldens            # check + (re)write fixtures

What it does (SURVEY.md §8c recipe):
  1. imports the reference's model/DCNet_model.py, model/test_DCNet_model.py and
     train_DCNet.py from /root/reference with stub modules for the packages the
     image lacks (cv2, torchvision, pytorch_pretrained_bert) — stubs are empty
     shells for *imports the hot path never calls*, not stand-ins for its code;
  2. builds the reference model, loads the version-stable synthetic state_dict
     (dcnet_amd/utils/synth.py), runs eval and train(p_dropout=0) forwards, the
     five losses and backward on seeded inputs;
  3. asserts that oracle/dcnet_oracle.py + oracle/train_oracle.py reproduce every
     output (<=1e-5 abs on O(1) values; exact on sampled indices);
  4. stores the reference's outputs as small fixtures (tests/golden/).  A fixture
     is data only: seeds, shapes, outputs.  No reference source is stored.
"""
from __future__ import annotations

import collections
import collections.abc
import json
import os
import random
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")


def _stub_modules():
    collections.Iterable = collections.abc.Iterable          # utils/transforms.py:10
    def mod(name, **attrs):
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m
    mod("cv2", setNumThreads=lambda n: None)
    ppb = mod("pytorch_pretrained_bert"); ppb.__path__ = []
    mod("pytorch_pretrained_bert.tokenization", BertTokenizer=object)
    mod("pytorch_pretrained_bert.modeling", BertModel=object)
    tv = mod("torchvision"); tv.__path__ = []
    class _T:  # transforms shells (train_DCNet.py:420-425 builds them at main(), never here)
        def __init__(self, *a, **k): pass
    mod("torchvision.transforms", Compose=_T, ToTensor=_T, Normalize=_T)
    mod("torchvision.datasets"); mod("torchvision.models")
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.datasets = sys.modules["torchvision.datasets"]; tv.models = sys.modules["torchvision.models"]
    torch.Tensor.cuda = lambda self, *a, **k: self           # generate_coord's .cuda() (F6)
    torch.nn.Module.cuda = lambda self, *a, **k: self


def _import_reference():
    _stub_modules()
    os.chdir(REF)
    # our repo also has a top-level ``model`` package (the drop-in); make sure the
    # reference's wins inside this process
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    # the reference's model/ has no __init__.py (namespace package) and would lose against our own
    # regular ``model`` package whatever the path order: bind the name to the reference explicitly
    pkg = types.ModuleType("model"); pkg.__path__ = [os.path.join(REF, "model")]; pkg.__package__ = "model"
    sys.modules["model"] = pkg
    import model.darknet as rdark
    assert rdark.__file__.startswith(REF), rdark.__file__
    rdark.Darknet._real_load_weights = rdark.Darknet.load_weights   # kept for oracle/make_format_goldens.py
    rdark.Darknet.load_weights = lambda self, p: None        # saved_models is a dangling symlink (F6)
    return rdark


def _ref_model_module(which: str, P: int):
    """exec the reference model file with the literal 1344 replaced by P (F3)."""
    fn = os.path.join(REF, "model", which + ".py")
    src = open(fn).read().replace("1344", str(P))
    m = types.ModuleType("model." + which + f"_P{P}")
    m.__package__ = "model"; m.__file__ = fn
    exec(compile(src, fn, "exec"), m.__dict__)
    return m


def _P(size):
    return sum((size // 32 * 2 ** i) ** 2 for i in range(3))


def _zero_dropout(model):
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def _maxdiff(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())


def _sample(t: torch.Tensor, n: int = 4096) -> np.ndarray:
    """Deterministic strided sample of a big tensor (fixtures stay small)."""
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.numpy().copy()
    step = f.numel() // n
    return f[::step][:n].numpy().copy()


def _stats(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double()
    return np.array([d.sum().item(), d.abs().sum().item(), (d * d).sum().item()])


def main():
    torch.manual_seed(0)
    rdark = _import_reference()
    sys.path.insert(0, ROOT)
    from dcnet_amd.utils.synth import synth_state_dict, synth_inputs, synth_boxes, apply_bn_calibration
    from oracle import dcnet_oracle as O
    from oracle import train_oracle as TO

    # ---- 0. graph check: generated defs == parsed cfg, slot by slot -----------------
    defs_ref = rdark.parse_model_config(os.path.join(REF, "model", "yolov3.cfg"))[1:]
    defs = O.darknet_defs()
    assert len(defs_ref) == len(defs) == 107
    for i, (a, b) in enumerate(zip(defs_ref, defs)):
        ta = {"convolutional": "conv", "yoloconvolutional": "yoloconv"}.get(a["type"], a["type"])
        assert ta == b["type"], (i, a, b)
        if ta in ("conv", "yoloconv"):
            assert (int(a["filters"]), int(a["size"]), int(a["stride"]), bool(int(a["batch_normalize"])),
                    a["activation"] == "leaky") == (b["filters"], b["size"], b["stride"], b["bn"], b["leaky"]), (i, a, b)
        elif ta == "shortcut":
            assert int(a["from"]) == b["frm"]
        elif ta == "route":
            assert [int(x) for x in a["layers"].split(",")] == b["layers"]
    print("graph: 107 slots match model/yolov3.cfg")

    corpus = list(range(1000))
    keys_written = False

    # ---- 0b. BatchNorm calibration: with random weights and default running stats the
    # eval-mode residual trunk doubles its variance every block (2^23).  One train-mode
    # pass of the REFERENCE with momentum=1 records batch statistics as running stats;
    # they ship as a 0.3 MB fixture and are applied on top of the synthetic weights.
    calib_path = os.path.join(GOLD, "bn_calib.npz")
    cm = _ref_model_module("DCNet_model", _P(256)).grounding_model(
        corpus=corpus, light=False, emb_size=512, coordmap=True, bert_model="bert-base-uncased", dataset="vid")
    cshapes = {k: tuple(v.shape) for k, v in cm.state_dict().items()}
    cm.load_state_dict(synth_state_dict(cshapes, seed=0), strict=True)
    _zero_dropout(cm)
    for mod in cm.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.momentum = 1.0
    cm.train()
    ci, cw, cwm = synth_inputs(4, 256, seed=999)
    random.seed(1)
    with torch.no_grad():
        cm(ci, cw, cwm)
    calib = {k: v.numpy().astype(np.float32) for k, v in cm.state_dict().items()
             if k.endswith(("running_mean", "running_var"))}
    np.savez_compressed(calib_path, **calib)
    print(f"bn calibration: {len(calib)} buffers -> {calib_path}")
    del cm
    report = {}
    only = os.environ.get("DCN_GOLDEN_ONLY")          # e.g. "416,4": (re)generate one pair-model case, keep the other fixtures
    if only:
        with open(os.path.join(GOLD, "PIN_REPORT.json")) as f:
            report = json.load(f)
    for size, N in ((256, 2), (416, 2), (256, 4), (416, 4)):
        if only and only != f"{size},{N}":
            continue
        P = _P(size)
        ref_train_mod = _ref_model_module("DCNet_model", P)
        model = ref_train_mod.grounding_model(corpus=corpus, light=False, emb_size=512, coordmap=True,
                                              bert_model="bert-base-uncased", dataset="vid")
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        if not keys_written and size == 256:
            assert len(shapes) == 597
            with open(os.path.join(GOLD, "state_dict_keys_256.json"), "w") as f:
                json.dump({k: list(s) for k, s in shapes.items()}, f)
            keys_written = True
        sd = apply_bn_calibration(synth_state_dict(shapes, seed=0), calib_path)
        model.load_state_dict(sd, strict=True)
        _zero_dropout(model)
        image, word_id, word_mask = synth_inputs(N, size, seed=size + N,
                                                 n_words=10 if (size, N) == (416, 2) else None)      # (416, 4): the benchmark geometry, train mode (round-3: verdict item 9)
        tag = f"S{size}_N{N}"

        # ---- 1. eval forward (4-tuple) ------------------------------------------------
        model.eval()
        random.seed(13)
        with torch.no_grad():
            outbox, sim, loc, only_obj = model(image, word_id, word_mask)
        sdo = {k: v.clone() for k, v in sd.items()}
        random.seed(13)
        with torch.no_grad():
            o = O.grounding_forward_pairs(sdo, image, word_id, training=False)
        d = max(max(_maxdiff(a, b) for a, b in zip(outbox, o["outbox"])),
                max(_maxdiff(a, b) for a, b in zip(sim, o["sim_score"])),
                max(_maxdiff(a, b) for a, b in zip(only_obj, o["only_obj"])))
        dloc = max(_maxdiff(a, b) for a, b in zip(loc, o["loc_score"]))
        print(f"{tag} eval: max|ref-oracle| = {d:.3e} (loc_score, min-max normalised: {dloc:.3e})")
        # (416^2 with N = 4: the reference's own CPU convolutions take another blocking than the oracle's calls at this batch
        #  size: 3e-5 absolute on outbox values of magnitude ~10, i.e. 3e-6 relative — still two fp32 runs of the same maths)
        assert d < (5e-5 if (size, N) == (416, 4) else 1e-5) and dloc < 2e-4, (d, dloc)
        # backbone taps straight from the reference Darknet
        with torch.no_grad():
            taps = model.visumodel(image)
        dt = max(_maxdiff(a, b) for a, b in zip(taps, o["taps"]))
        assert dt < 1e-4, dt
        boxes = O.decode_boxes([x.clone() for x in outbox], size)
        gold = dict(size=size, n=N, seed_inputs=size + N,
                    boxes=boxes.numpy())
        for s in range(3):
            gold[f"outbox{s}"] = outbox[s].numpy(); gold[f"sim{s}"] = sim[s].numpy()
            gold[f"loc{s}"] = loc[s].numpy(); gold[f"only_obj{s}"] = only_obj[s].numpy()
            gold[f"tap{s}_sample"] = _sample(taps[s]); gold[f"tap{s}_stats"] = _stats(taps[s])
        np.savez_compressed(os.path.join(GOLD, f"eval_{tag}.npz"), **gold)
        report[tag + "_eval"] = d

        # ---- 2. train forward (11-tuple), losses, backward ---------------------------
        if (size, N) == (416, 2):
            continue        # 416 train goldens: eval covers the shapes; keep CPU suite short
        model.load_state_dict(sd, strict=True)
        model.train()
        import train_DCNet as T
        T.args = SimpleNamespace(size=size, anchor_imsize=416)
        T.anchors_full = list(O.ANCHORS_FULL)
        bbox = synth_boxes(N, size, seed=size + N)
        random.seed(13)
        outs = model(image, word_id, word_mask)
        (pred, sim, loc, fv, fa, ff, cf, nf, vp, lp, nc) = outs
        bb = torch.clamp(bbox, min=0, max=size - 1)
        gt_param, gi, gj, best_n, gt_center = T.build_target(bb, pred)
        pred5 = [p.view(p.size(0), 3, 5, p.size(2), p.size(3)) for p in pred]
        neg_sim = [torch.sum(fa[range(fa.size(0) - 1, -1, -1), :, :, :] * fv[ii][:, :512], dim=1) for ii in range(3)]
        l_yolo = T.yolo_loss(pred5, gt_param, gi, gj, best_n)
        l_rank = T.rank_loss(sim, neg_sim, gt_center, gi, gj, best_n, w_coord=0.)
        l_inter = T.Interframe_contrastive_loss(ff, cf, nf)
        l_cross = T.Crossmodal_constrastive_loss(vp, lp, nc)
        l_loc = T.loc_loss(loc, sim, gt_center)
        loss = l_yolo + 100 * l_rank + l_loc + 100 * l_inter + l_cross
        model.zero_grad()
        loss.backward()
        ref_losses = dict(yolo=l_yolo, rank=l_rank, interframe=l_inter, cross=l_cross, loc=l_loc)
        ref_grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        ref_nograd = sorted(k for k, p in model.named_parameters() if p.grad is None)
        ref_buffers = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}

        sdo = {k: v.clone() for k, v in sd.items()}
        params = {k: sdo[k].requires_grad_(True) for k, _ in model.named_parameters()}
        random.seed(13)
        o = O.grounding_forward_pairs(sdo, image, word_id, training=True, skip_dead=False)
        oloss, olosses = TO.total_loss(o, bbox, size)
        oloss.backward()
        for k in ref_losses:
            dl = abs(float(ref_losses[k]) - float(olosses[k]))
            print(f"{tag} train loss {k}: ref {float(ref_losses[k]):.6f} oracle {float(olosses[k]):.6f}")
            assert dl < 2e-5 * max(1.0, abs(float(ref_losses[k]))), (k, dl)
        d_out = max(max(_maxdiff(a, b) for a, b in zip(pred, o["outbox"])),
                    max(_maxdiff(a, b) for a, b in zip(fv, o["corr_feat"])),
                    _maxdiff(fa, o["flang_attn"]),
                    max(_maxdiff(a, b) for a, b in zip(ff, o["frame_feature"])),
                    max(_maxdiff(a, b) for a, b in zip(cf, o["corrspendence_feature"])),
                    max(_maxdiff(a, b) for a, b in zip(nf, o["neg_feature"])),
                    max(_maxdiff(a, b) for a, b in zip(vp, o["vit_posit"])),
                    max(_maxdiff(a, b) for a, b in zip(lp, o["lag_posit"])),
                    max(_maxdiff(a, b) for a, b in zip(nc, o["neg_cross"])))
        print(f"{tag} train: max|ref-oracle| over the 11 outputs = {d_out:.3e}")
        assert d_out < 1e-4, d_out   # N=2 train-mode BN1d amplifies 1e-7 LSTM rounding
        # NOTE on conditioning: at N=2..4 the reference's own gradient moves by 0.2-4 % on
        # the scale-2 branch when an input is perturbed by 1e-7 relative (min-max
        # normalised loc_score + tiny-batch BN), so gradients are pinned to 5e-2 worst /
        # 2e-3 median, the forward outputs to 1e-4, the loss scalars to 2e-5.
        worst = 0.0; rels = []
        for k, g in ref_grads.items():
            og = params[k].grad
            assert og is not None, k
            if float(g.abs().max()) < 1e-3 or k in ("mapping_lang.0.bias", "mapping_lang.4.bias",
                                                    "loc_embedding.0.bias", "loc_text_embedding.0.bias"):
                continue     # a bias feeding a train-mode BN: gradient is 0 up to rounding
            rel = _maxdiff(g, og) / float(g.abs().max())
            rels.append(rel)
            worst = max(worst, rel)
        med = float(np.median(rels))
        print(f"{tag} train: relative grad diff worst {worst:.3e} median {med:.3e} over {len(rels)} params; "
              f"params without grad: {len(ref_nograd)}")
        assert worst < 5e-2 and med < 2e-3, (worst, med)
        for k, v in ref_buffers.items():
            assert _maxdiff(v, sdo[k]) < 1e-5 * max(1.0, float(v.abs().max())), k
        gold = dict(size=size, n=N, seed_inputs=size + N,
                    losses=np.array([float(ref_losses[k]) for k in ("yolo", "rank", "interframe", "cross", "loc")]),
                    k9_q=o["k9_idx"]["q"].numpy(), k9_k=o["k9_idx"]["k"].numpy(), k9_neg=o["k9_idx"]["neg"].numpy(),
                    k14_word=o["k14_idx"]["word"].numpy(), k14_neg=o["k14_idx"]["neg"].numpy(),
                    nograd=np.array(ref_nograd))
        for s in range(3):
            gold[f"outbox{s}"] = pred[s].detach().numpy(); gold[f"sim{s}"] = sim[s].detach().numpy()
            gold[f"loc{s}"] = loc[s].detach().numpy()
            gold[f"corr_feat{s}_sample"] = _sample(fv[s]); gold[f"corr_feat{s}_stats"] = _stats(fv[s])
        gkeys = ["visumodel.module_list.0.conv_0.weight", "visumodel.module_list.0.batch_norm_0.weight",
                 "visumodel.module_list.42.conv_42.weight", "visumodel.module_list.78.conv_78.weight",
                 "visumodel.module_list.102.conv_102.weight", "mapping_visu.2.conv.weight",
                 "corr_conv.0.0.conv.weight", "fcn_emb.1.1.conv.weight", "fcn_out.0.1.weight",
                 "textmodel.rnn.weight_hh_l0", "textmodel.embedding.weight", "sub_attn.fc.weight",
                 "loc_text_embedding.0.weight", "loc_embedding.0.weight", "mapping_lang.4.weight"]
        gold["grad_keys"] = np.array(gkeys)
        gold["grad_norms"] = np.array([float(ref_grads[k].double().norm()) for k in gkeys])
        for k in ("visumodel.module_list.0.conv_0.weight", "fcn_out.0.1.weight"):
            gold["grad::" + k] = ref_grads[k].numpy()
        gold["bn_rm::visumodel.module_list.0.batch_norm_0"] = ref_buffers["visumodel.module_list.0.batch_norm_0.running_mean"].numpy()
        gold["bn_rv::mapping_visu.0.bn"] = ref_buffers["mapping_visu.0.bn.running_var"].numpy()
        np.savez_compressed(os.path.join(GOLD, f"train_{tag}.npz"), **gold)
        report[tag + "_train_out"] = d_out; report[tag + "_train_grad_rel_worst"] = worst; report[tag + "_train_grad_rel_median"] = med

    # ---- 3. n_frame inference model (model/test_DCNet_model.py) -----------------------
    for size, B, T_ in (() if only else ((256, 1, 5), (256, 2, 2), (416, 1, 8))):
        P = _P(size)
        tm = _ref_model_module("test_DCNet_model", P)
        model = tm.grounding_model(corpus=corpus, light=False, emb_size=512, coordmap=True,
                                   bert_model="bert-base-uncased", dataset="vid")
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = apply_bn_calibration(synth_state_dict(shapes, seed=0), calib_path)
        model.load_state_dict(sd, strict=True)
        model.eval()
        image, word_id, word_mask = synth_inputs(B * T_, size, n_queries=B, seed=size + 7 * T_)
        with torch.no_grad():
            outbox, sim, loc, corr, only_obj = model(image, word_id, word_mask, T_)
            o = O.grounding_forward_nframe({k: v.clone() for k, v in sd.items()}, image, word_id, T_)
        d = max(max(_maxdiff(a, b) for a, b in zip(outbox, o["outbox"])),
                max(_maxdiff(a, b) for a, b in zip(sim, o["sim_score"])),
                max(_maxdiff(a, b) for a, b in zip(corr, o["corr_feat"])),
                max(_maxdiff(a, b) for a, b in zip(only_obj, o["only_obj"])))
        dloc = max(_maxdiff(a, b) for a, b in zip(loc, o["loc_score"]))
        tag = f"S{size}_B{B}_T{T_}"
        print(f"{tag} n_frame eval: max|ref-oracle| = {d:.3e} (loc {dloc:.3e})")
        assert d < 2e-5 and dloc < 2e-4, (d, dloc)
        gold = dict(size=size, b=B, t=T_, seed_inputs=size + 7 * T_,
                    boxes=O.decode_boxes([x.clone() for x in outbox], size).numpy())
        for s in range(3):
            gold[f"outbox{s}"] = outbox[s].numpy(); gold[f"sim{s}"] = sim[s].numpy()
            gold[f"loc{s}"] = loc[s].numpy(); gold[f"only_obj{s}"] = only_obj[s].numpy()
            gold[f"corr_feat{s}_sample"] = _sample(corr[s]); gold[f"corr_feat{s}_stats"] = _stats(corr[s])
        np.savez_compressed(os.path.join(GOLD, f"nframe_{tag}.npz"), **gold)
        report[tag + "_nframe"] = d
    with open(os.path.join(GOLD, "PIN_REPORT.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("goldens written to", GOLD)


if __name__ == "__main__":
    main()
