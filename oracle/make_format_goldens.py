"""Pin the caller-side decode and the two on-disk formats to the REAL reference.

Runs ONLY in the build container (needs /root/reference, never the GPU box):

    python -m oracle.make_format_goldens

TEST INFRASTRUCTURE (see oracle/dcnet_oracle.py header).  Writes data-only fixtures under tests/golden/:

  decode_ref.npz      inputs (seeded outbox tensors, GT boxes) and what the reference's own ``validate_epoch``
                      (train_DCNet.py:728-846) made of them: decoded boxes (:764-810), IoU and Acc@0.5 (:813-816).
                      The loop is driven with a one-batch loader and a model shell that returns the stored tensors;
                      ``bbox_iou`` is wrapped to capture ``pred_bbox``.  Also asserts oracle.decode_boxes == reference.
  formats_ref.json    * darknet binary: the product's ``Darknet.save_weights`` file is read by the reference's
                        ``Darknet.load_weights`` (model/darknet.py:433-483) and every tensor compared; the reference's
                        ``save_weights`` (:490-513) file is compared byte for byte with the product's
                        ``save_weights(reference_layout=True)``.  Stored: file sizes and sha256 digests.
                      * ``.pth.tar``: a checkpoint written by the reference's ``save_checkpoint`` (train_DCNet.py:255-263)
                        from a DDP-style ``module.``-prefixed state_dict and the reference's two-group RMSprop
                        (:519-534), loaded by ``dcnet_amd.train.load_checkpoint`` into the product model + optimizer.
                        Stored: key list digest, group sizes, per-tensor crc32 of a few entries.
"""
from __future__ import annotations

import hashlib
import json
import os
import random
import sys
import tempfile
import zlib
from types import SimpleNamespace

import numpy as np
import torch

from .make_goldens import GOLD, REF, ROOT, _P, _import_reference, _ref_model_module


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def _crc(t: torch.Tensor) -> int:
    return zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())


def _fake_outbox(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, 15, x, x, generator=g) for x in (size // 32, size // 16, size // 8)]


def main():
    rdark = _import_reference()
    sys.path.insert(0, ROOT)
    from dcnet_amd.utils.synth import synth_boxes, synth_state_dict
    from oracle import dcnet_oracle as O
    import train_DCNet as T
    import logging
    logging.disable(logging.CRITICAL)

    # ---- 1. decode + Acc@0.5 through the reference's validate_epoch ----------------------------------------
    gold = {}
    for size in (256, 416):
        n = 5
        outbox = _fake_outbox(n, size, 3)
        # GT boxes = the decoded boxes shifted by 0 / 5 / 20 / 60 / 200 % of their width, so that Acc@0.5 has hits and misses
        pre = O.decode_boxes([o.clone() for o in outbox], size)
        bbox = pre + (pre[:, 2:3] - pre[:, 0:1]) * torch.tensor([0., 0.05, 0.2, 0.6, 2.0]).view(n, 1)
        T.args = SimpleNamespace(size=size, anchor_imsize=416, save_plot=False, dataset="vid")
        T.anchors_full = list(O.ANCHORS_FULL)
        cap = {}
        real_iou = T.bbox_iou

        def spy(b1, b2, x1y1x2y2=True):
            cap["pred"], cap["gt"] = b1.clone(), b2.clone()
            cap["iou"] = real_iou(b1, b2, x1y1x2y2=x1y1x2y2)
            return cap["iou"]

        class Shell:
            def eval(self): return self
            def __call__(self, image, word_id, word_mask):
                return [o.clone() for o in outbox], None, None, None

        T.bbox_iou = spy
        try:
            loader = [(torch.zeros(n, 3, 8, 8), torch.ones(n, 20, dtype=torch.long), torch.zeros(n, 20, dtype=torch.long), bbox, None)]
            acc = T.validate_epoch(loader, Shell(), True)
        finally:
            T.bbox_iou = real_iou
        mine = O.decode_boxes([o.clone() for o in outbox], size)
        d = float((mine - cap["pred"]).abs().max())
        print(f"decode {size}: max|oracle - reference| = {d:.2e} px, reference Acc@0.5 = {acc:.3f}")
        assert d < 1e-4, d
        iou_o = O.bbox_iou_xyxy(mine, torch.clamp(bbox, 0, size - 1))
        assert float((iou_o - cap["iou"]).abs().max()) < 1e-6
        for s in range(3):
            gold[f"outbox{s}_{size}"] = outbox[s].numpy()
        gold[f"pred_bbox_{size}"] = cap["pred"].numpy(); gold[f"gt_bbox_{size}"] = cap["gt"].numpy()
        gold[f"iou_{size}"] = cap["iou"].numpy(); gold[f"accu_{size}"] = np.float64(acc)
    np.savez_compressed(os.path.join(GOLD, "decode_ref.npz"), **gold)

    # ---- 2. darknet binary weights -----------------------------------------------------------------------
    from dcnet_amd.darknet import Darknet as PDarknet
    with open(os.path.join(GOLD, "state_dict_keys_256.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = synth_state_dict(shapes, seed=0)
    vsd = {k[len("visumodel."):]: v for k, v in sd.items() if k.startswith("visumodel.")}
    rep = {}
    with tempfile.TemporaryDirectory() as tmp:
        prod = PDarknet(config_path=os.path.join(ROOT, "model", "yolov3.cfg"))
        prod.load_state_dict(vsd, strict=True)
        prod.seen = 4321
        p_all = os.path.join(tmp, "product_all.weights")
        prod.save_weights(p_all)
        ref = rdark.Darknet(config_path=os.path.join(REF, "model", "yolov3.cfg"))
        rdark.Darknet._real_load_weights(ref, p_all)          # the reference's own reader (make_goldens patches the attribute)
        assert int(ref.seen) == 4321
        rsd = ref.state_dict()
        for k, v in vsd.items():
            if "num_batches_tracked" not in k:
                assert torch.equal(rsd[k], v), k
        rep["weights_product_file_read_by_reference"] = {"bytes": os.path.getsize(p_all), "sha256": _sha(p_all), "seen": 4321,
                                                         "tensors_compared": sum("num_batches" not in k for k in vsd)}
        r_file = os.path.join(tmp, "reference.weights")
        ref.save_weights(r_file)
        p_ref = os.path.join(tmp, "product_reflayout.weights")
        prod.save_weights(p_ref, reference_layout=True)
        assert _sha(r_file) == _sha(p_ref), "product's reference_layout file differs from the reference writer's"
        rep["weights_reference_writer_file"] = {"bytes": os.path.getsize(r_file), "sha256": _sha(r_file)}
        print("darknet weights:", rep["weights_product_file_read_by_reference"]["bytes"], "bytes read back by the reference;",
              "reference writer's file", rep["weights_reference_writer_file"]["bytes"], "bytes == product reference_layout")

        # ---- 3. .pth.tar checkpoint ----------------------------------------------------------------------
        model = _ref_model_module("DCNet_model", _P(256)).grounding_model(
            corpus=list(range(1000)), light=False, emb_size=512, coordmap=True, bert_model="bert-base-uncased", dataset="vid")
        model.load_state_dict(sd, strict=True)
        wrapped = torch.nn.Sequential()
        wrapped.add_module("module", model)                    # DDP-style 'module.' key prefix (train_DCNet.py:483,553)
        visu_param = list(model.visumodel.parameters())
        ids = {id(p) for p in visu_param}
        rest_param = [p for p in wrapped.parameters() if id(p) not in ids]          # train_DCNet.py:519-523
        opt = torch.optim.RMSprop([{"params": rest_param}, {"params": visu_param, "lr": 1e-5}], lr=1e-4, weight_decay=0.0005)
        # one fake step so that the optimizer carries state for every parameter that gets a gradient in the reference
        gold_t = np.load(os.path.join(GOLD, "train_S256_N4.npz"), allow_pickle=True)
        nograd = {"module." + str(k) for k in gold_t["nograd"]}
        g = torch.Generator().manual_seed(5)
        for k, p in wrapped.named_parameters():
            if k not in nograd:
                p.grad = torch.randn(p.shape, generator=g) * 1e-3
        lrs = [g_["lr"] for g_ in opt.param_groups]
        for g_ in opt.param_groups:
            g_["lr"] = 0.0                                     # state gets filled, weights stay the synthetic ones
        opt.step()
        for g_, lr_ in zip(opt.param_groups, lrs):
            g_["lr"] = lr_
        T.args = SimpleNamespace(dataset="vid", batch_size=8)
        os.makedirs(os.path.join(tmp, "saved_models"))
        cwd = os.getcwd(); os.chdir(tmp)
        try:
            T.save_checkpoint({"epoch": 7, "state_dict": wrapped.state_dict(), "best_loss": 0.125, "optimizer": opt.state_dict()},
                              True, "pin")
        finally:
            os.chdir(cwd)
        ck = os.path.join(tmp, "saved_models", "pin_checkpoint.pth.tar")
        assert os.path.exists(ck) and os.path.exists(os.path.join(tmp, "saved_models", "pin_model_best.pth.tar"))
        # product side, CPU objects only (no kernels involved): model + optimizer of dcnet_amd.train
        from dcnet_amd import train as PT
        from dcnet_amd.model import grounding_model as PModel
        from dcnet_amd.parallel import freeze_gradless
        pm = PModel(corpus=list(range(1000)), emb_size=512, img_size=256, weights_path=None,
                    config_path=os.path.join(ROOT, "model", "yolov3.cfg"))
        freeze_gradless(pm)
        popt = PT.make_optimizer(pm, 1e-4)
        epoch, best = PT.load_checkpoint(pm, ck, popt)
        assert (epoch, best) == (7, 0.125)
        psd = pm.state_dict()
        for k, v in wrapped.state_dict().items():
            assert torch.equal(psd[k[7:]], v), k
        assert [len(g_["params"]) for g_ in popt.state_dict()["param_groups"]] == [93, 222]
        ref_state = opt.state_dict()["state"]; got_state = popt.state_dict()["state"]
        assert sorted(ref_state) == sorted(got_state)
        for i in ref_state:
            assert torch.equal(ref_state[i]["square_avg"], got_state[i]["square_avg"]), i
        keys = list(wrapped.state_dict().keys())
        rep["checkpoint_reference_writer"] = {
            "epoch": 7, "best_loss": 0.125, "n_keys": len(keys),
            "keys_sha256": hashlib.sha256("\n".join(keys).encode()).hexdigest(),
            "optimizer_group_sizes": [len(g_["params"]) for g_ in opt.state_dict()["param_groups"]],
            "optimizer_state_entries": len(ref_state),
            "crc32": {k: _crc(wrapped.state_dict()[k]) for k in ("module.visumodel.module_list.0.conv_0.weight",
                                                                  "module.textmodel.rnn.weight_hh_l0", "module.fcn_out.2.1.bias")},
            "square_avg_crc32": {str(i): _crc(ref_state[i]["square_avg"]) for i in (0, 92, 93, 314) if i in ref_state}}
        print("checkpoint: reference save_checkpoint -> product load_checkpoint ok;", rep["checkpoint_reference_writer"]["optimizer_group_sizes"],
              "groups,", len(ref_state), "state entries")
    # ---- 4. light=True head (model/DCNet_model.py:296-312) -------------------------------------------------
    from dcnet_amd.utils.synth import apply_bn_calibration, synth_inputs
    lm = _ref_model_module("DCNet_model", _P(256)).grounding_model(
        corpus=list(range(1000)), light=True, emb_size=512, coordmap=True, bert_model="bert-base-uncased", dataset="vid")
    lshapes = {k: tuple(v.shape) for k, v in lm.state_dict().items()}
    lsd = apply_bn_calibration(synth_state_dict(lshapes, seed=0), os.path.join(GOLD, "bn_calib.npz"))
    lm.load_state_dict(lsd, strict=True)
    lm.eval()
    image, word_id, word_mask = synth_inputs(2, 256, seed=77)
    random.seed(13)
    with torch.no_grad():
        outbox, sim, loc, only_obj = lm(image, word_id, word_mask)
        o = O.grounding_forward_pairs({k: v.clone() for k, v in lsd.items()}, image, word_id, training=False, sample=False)
    d = max(max(float((a - b).abs().max()) for a, b in zip(outbox, o["outbox"])), max(float((a - b).abs().max()) for a, b in zip(sim, o["sim_score"])))
    dl = max(float((a - b).abs().max()) for a, b in zip(loc, o["loc_score"]))
    print(f"light=True eval 256 N=2: max|ref-oracle| = {d:.2e} (loc {dl:.2e}); {len(lshapes)} state_dict keys")
    assert d < 1e-5 and dl < 2e-4
    lg = dict(size=256, n=2, seed_inputs=77, keys=np.array(list(lshapes.keys())))
    for s_ in range(3):
        lg[f"outbox{s_}"] = outbox[s_].numpy(); lg[f"sim{s_}"] = sim[s_].numpy(); lg[f"loc{s_}"] = loc[s_].numpy()
    np.savez_compressed(os.path.join(GOLD, "eval_light_S256_N2.npz"), **lg)
    rep["light_model"] = {"n_keys": len(lshapes), "n_params": int(sum(p.numel() for p in lm.parameters()))}
    with open(os.path.join(GOLD, "formats_ref.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print("format fixtures written to", GOLD)


if __name__ == "__main__":
    main()
