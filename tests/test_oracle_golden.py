"""CPU: the oracle (oracle/) re-checked against the fixtures captured from the REAL reference
(tests/golden, written by oracle/make_goldens.py in the build container).  This is the pin that
travels: it runs anywhere, without /root/reference."""
import json
import os
import random

import numpy as np
import pytest
import torch

from util import GOLD, maxdiff, synth_sd


def _inputs(gold):
    from dcnet_amd.utils.synth import synth_inputs
    size, n = int(gold["size"]), int(gold["n"])
    return size, n, synth_inputs(n, size, seed=int(gold["seed_inputs"]), n_words=10 if (size, n) == (416, 2) else None)


@pytest.mark.parametrize("tag", ["S256_N2", "S416_N2"])
def test_oracle_eval_matches_reference_outputs(tag):
    from oracle import dcnet_oracle as O
    gold = np.load(os.path.join(GOLD, f"eval_{tag}.npz"))
    size, n, (image, word_id, _) = _inputs(gold)
    with torch.no_grad():
        o = O.grounding_forward_pairs(synth_sd(size), image, word_id, training=False, sample=False)
    for s in range(3):
        assert maxdiff(o["outbox"][s], torch.from_numpy(gold[f"outbox{s}"])) < 2e-5
        assert maxdiff(o["sim_score"][s], torch.from_numpy(gold[f"sim{s}"])) < 2e-5
        assert maxdiff(o["loc_score"][s], torch.from_numpy(gold[f"loc{s}"])) < 2e-4      # min-max normalised
        assert maxdiff(o["only_obj"][s], torch.from_numpy(gold[f"only_obj{s}"])) < 2e-5
        st = gold[f"tap{s}_stats"]
        t = o["taps"][s].double()
        assert abs(float(t.abs().sum()) - st[1]) < 1e-4 * st[1]
    boxes = O.decode_boxes([x.clone() for x in o["outbox"]], size)
    assert maxdiff(boxes, torch.from_numpy(gold["boxes"])) < 1e-2


def test_oracle_nframe_matches_reference_outputs():
    from dcnet_amd.utils.synth import synth_inputs
    from oracle import dcnet_oracle as O
    gold = np.load(os.path.join(GOLD, "nframe_S256_B1_T5.npz"))
    size, b, t = int(gold["size"]), int(gold["b"]), int(gold["t"])
    image, word_id, _ = synth_inputs(b * t, size, n_queries=b, seed=int(gold["seed_inputs"]))
    with torch.no_grad():
        o = O.grounding_forward_nframe(synth_sd(size), image, word_id, t)
    for s in range(3):
        assert maxdiff(o["outbox"][s], torch.from_numpy(gold[f"outbox{s}"])) < 2e-5
        assert maxdiff(o["sim_score"][s], torch.from_numpy(gold[f"sim{s}"])) < 2e-5
        assert maxdiff(o["loc_score"][s], torch.from_numpy(gold[f"loc{s}"])) < 2e-4


@pytest.mark.parametrize("tag", ["S256_N4", "S416_N4"])
def test_oracle_train_losses_indices_and_grads_match_reference(tag):
    """S416_N4 = the benchmark geometry (416x416: 13/26/52 grids, P = 3549) in train mode, 4 images."""
    from dcnet_amd.utils.synth import synth_boxes
    from oracle import dcnet_oracle as O
    from oracle import train_oracle as TO
    gold = np.load(os.path.join(GOLD, f"train_{tag}.npz"), allow_pickle=True)
    size, n, (image, word_id, _) = _inputs(gold)
    sd = synth_sd(size)
    nograd = set(str(k) for k in gold["nograd"])
    params = {}
    for k, v in sd.items():
        if v.dtype == torch.float32 and "running" not in k:
            params[k] = v.requires_grad_(True)
    random.seed(13)
    o = O.grounding_forward_pairs(sd, image, word_id, training=True, skip_dead=False)
    loss, parts = TO.total_loss(o, synth_boxes(n, size, seed=int(gold["seed_inputs"])), size)
    loss.backward()
    for k, ref in zip(("yolo", "rank", "interframe", "cross", "loc"), gold["losses"]):
        assert abs(float(parts[k]) - float(ref)) < 2e-5 * max(1.0, abs(float(ref))), k
    assert np.array_equal(o["k9_idx"]["q"].numpy(), gold["k9_q"]) and np.array_equal(o["k9_idx"]["k"].numpy(), gold["k9_k"])
    assert np.array_equal(o["k9_idx"]["neg"].numpy(), gold["k9_neg"])
    assert np.array_equal(o["k14_idx"]["word"].numpy(), gold["k14_word"]) and np.array_equal(o["k14_idx"]["neg"].numpy(), gold["k14_neg"])
    for k, ref in zip(gold["grad_keys"], gold["grad_norms"]):
        g = params[str(k)].grad
        assert g is not None and abs(float(g.double().norm()) - float(ref)) < 1e-3 * float(ref), k
    for k in ("visumodel.module_list.0.conv_0.weight", "fcn_out.0.1.weight"):
        ref = torch.from_numpy(gold["grad::" + k])
        assert maxdiff(params[k].grad, ref) < 1e-3 * float(ref.abs().max())
    got_nograd = {k for k, p in params.items() if p.grad is None}
    assert got_nograd == nograd, (sorted(got_nograd ^ nograd))


def test_pin_report_is_within_bounds():
    with open(os.path.join(GOLD, "PIN_REPORT.json")) as f:
        rep = json.load(f)
    assert len(rep) >= 10
    for k, v in rep.items():
        if k.endswith("_eval") or k.endswith("_nframe") or k.endswith("_train_out"):
            assert v < 1e-4, (k, v)


def test_oracle_light_head_matches_reference_fixture():
    """light=True (model/DCNet_model.py:296-312): one-block fcn_emb, bare Conv2d fcn_out; fixture written by
    oracle/make_format_goldens.py from the real reference."""
    import numpy as np
    import os
    from dcnet_amd.utils.synth import apply_bn_calibration, synth_inputs, synth_state_dict
    from oracle import dcnet_oracle as O
    from util import GOLD, ref_shapes
    g = np.load(os.path.join(GOLD, "eval_light_S256_N2.npz"), allow_pickle=True)
    full = ref_shapes(256)
    shapes = {}
    for k in (str(x) for x in g["keys"]):
        if k in full:
            shapes[k] = full[k]
        elif k.startswith("fcn_out."):                       # fcn_out.{s}.0.{weight,bias}: Conv2d(512, 15, 1)
            shapes[k] = (15, 512, 1, 1) if k.endswith("weight") else (15,)
    assert len(shapes) == len(g["keys"]) == 543
    sd = apply_bn_calibration(synth_state_dict(shapes, seed=0), os.path.join(GOLD, "bn_calib.npz"))
    image, word_id, _ = synth_inputs(2, 256, seed=77)
    with torch.no_grad():
        o = O.grounding_forward_pairs(sd, image, word_id, training=False, sample=False)
    for s in range(3):
        assert float((o["outbox"][s] - torch.from_numpy(g[f"outbox{s}"])).abs().max()) < 2e-5
        assert float((o["sim_score"][s] - torch.from_numpy(g[f"sim{s}"])).abs().max()) < 2e-5
        assert float((o["loc_score"][s] - torch.from_numpy(g[f"loc{s}"])).abs().max()) < 5e-4
