"""CPU: host-side logic of the product — cfg/plan compiler, checkpoint key contract, darknet weight
file io, the native MT19937 sampler, sharding."""
import json
import os
import random

import numpy as np
import torch

from util import GOLD, ROOT, ref_shapes


def test_generated_cfg_roundtrip_and_plan():
    from dcnet_amd import darknet as D
    blocks = D.yolov3_blocks()
    parsed = D.parse_model_config(os.path.join(ROOT, "model", "yolov3.cfg"))
    assert len(parsed) == len(blocks) == 108
    for a, b in zip(parsed, blocks):
        assert a["type"] == b["type"]
        for k in ("filters", "size", "stride", "activation", "layers", "from"):
            assert str(a.get(k)) == str(b.get(k)), (a, b)
        assert int(a.get("batch_normalize", 0)) == int(b.get("batch_normalize", 0))
    plan, taps, ch = D.build_plan(parsed[1:], 3)
    assert taps == [78, 90, 102] and [ch[t] for t in taps] == [1024, 512, 256]
    convs = [op for op in plan if isinstance(op, D._ConvOp)]
    assert len(convs) == 68                                     # 75 convs - 7 in the dead YOLO heads
    assert sum(op.res is not None for op in convs) == 23        # every residual add is fused
    assert not any(op.slot in (80, 81, 92, 93, 103, 104, 105) for op in convs)
    live_flops = sum(2 * op.cout * op.cin * op.k * op.k * (416 // _stride(plan, op, D)) ** 2 for op in convs)
    assert abs(live_flops / 1e9 - 60.28) < 0.6                  # 60.28 GF/image @416 (live layers)


def _stride(plan, op, D):
    # cumulative output stride of a conv slot in the 416 graph
    table = {}
    s = 1
    for o in plan:
        if isinstance(o, D._ConvOp):
            src = table.get(o.src, 1)
            table[o.dst] = src * o.stride
        elif isinstance(o, D._UpCatOp):
            table[o.dst] = table[o.lat_src]
        else:
            table[o.dst] = table[o.src]
    return table[op.dst]


def test_state_dict_contract():
    from model.DCNet_model import grounding_model
    m = grounding_model(corpus=list(range(1000)), emb_size=512, img_size=256, weights_path=None,
                        config_path=os.path.join(ROOT, "model", "yolov3.cfg"))
    sd = m.state_dict()
    ref = ref_shapes(256)
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == ref[k] for k in ref)
    assert sum(p.numel() for p in m.parameters()) == 80770344
    m416 = grounding_model(corpus=list(range(1000)), emb_size=512, img_size=416, weights_path=None, config_path="")
    assert m416.loc_text_embedding[0].weight.shape == (512, 3549)
    # the gradient-less set is exactly the reference's (captured with the real model)
    from dcnet_amd.parallel import gradless_parameter_names
    gold = np.load(os.path.join(GOLD, "train_S256_N4.npz"), allow_pickle=True)
    assert sorted(gradless_parameter_names(m)) == sorted(str(k) for k in gold["nograd"])


def test_darknet_weights_file_roundtrip(tmp_path):
    from dcnet_amd.darknet import Darknet
    torch.manual_seed(0)
    a = Darknet(config_path="")
    for p in a.parameters():
        torch.nn.init.normal_(p, std=0.02)
    for b in a.buffers():
        if b.dtype == torch.float32:
            b.uniform_(0.5, 1.5)
    a.seen = 1234
    path = str(tmp_path / "w.weights")
    a.save_weights(path)
    assert os.path.getsize(path) == 248007048 or os.path.getsize(path) == 5 * 4 + 62001757 * 4
    b = Darknet(config_path="")
    b.load_weights(path)
    assert b.seen == 1234
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        if "num_batches_tracked" not in k:
            assert torch.equal(va, vb), k


def test_native_sampler_is_bit_exact_with_python_random():
    from dcnet_amd.lib import lib
    L = lib()
    for hw, top_k, neg_n, pairs in ((64, 30, 10, 3), (169, 30, 10, 2), (361, 30, 10, 1)):
        random.seed(13)
        kpos = np.array([[random.randrange(hw) for _ in range(top_k)] for _ in range(pairs)], dtype=np.int64)
        st = random.getstate()
        ref = np.zeros((pairs, top_k, neg_n), dtype=np.int64)
        for p in range(pairs):
            for j in range(top_k):
                lst = list(range(hw)); lst.remove(int(kpos[p, j]))
                ref[p, j] = random.sample(lst, neg_n)
        after = random.getstate()
        state = np.array(st[1], dtype=np.uint32); out = np.zeros_like(ref)
        L.mt_sample_interframe(state.ctypes.data, kpos.ctypes.data, pairs, top_k, hw, neg_n, out.ctypes.data)
        assert (out == ref).all() and tuple(int(x) for x in state) == after[1]
    # (2, 64): the two populations differ in bit length (generic path); 169/170/676: the block path and its five-survivor
    # shortcut; 22/23: the pool branch on one / both populations; k = 8, 9: the set-size rule for k > 5
    for n, rows, k in ((2, 64, 5), (3, 169, 5), (3, 170, 5), (2, 676, 5), (2, 22, 5), (2, 23, 5), (2, 100, 8), (2, 400, 9)):
        random.seed(5); st = random.getstate()
        ref = np.zeros((n, rows, k), dtype=np.int64)
        for ii in range(n):
            for jj in range(rows):
                for index in range(n):
                    lst = list(range(rows))
                    if index == ii:
                        lst.remove(jj)
                    s = random.sample(lst, k)
                ref[ii, jj] = s
        after = random.getstate()
        state = np.array(st[1], dtype=np.uint32); out = np.zeros_like(ref)
        L.mt_sample_crossmodal(state.ctypes.data, n, rows, k, out.ctypes.data)
        assert (out == ref).all() and tuple(int(x) for x in state) == after[1]


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    from dcnet_amd.parallel import shard_indices
    for n, w in ((64, 8), (10, 4), (7, 2), (3, 8)):
        for r in range(w):
            ref = list(DistributedSampler(range(n), num_replicas=w, rank=r, shuffle=False))
            assert shard_indices(n, r, w) == ref


def test_lr_schedule_and_optimizer_groups():
    from dcnet_amd import train as T
    from model.DCNet_model import grounding_model
    m = grounding_model(corpus=list(range(50)), emb_size=512, img_size=256, weights_path=None, config_path="")
    from dcnet_amd.parallel import freeze_gradless
    freeze_gradless(m)
    opt = T.make_optimizer(m, 1e-4)
    assert len(opt.param_groups) == 2 and opt.param_groups[1]["lr"] == 1e-5 and opt.param_groups[0]["weight_decay"] == 0.0005
    # the reference's groups hold every parameter, trainable or not ([93, 222] tensors: oracle/make_format_goldens.py)
    assert [len(g["params"]) for g in opt.param_groups] == [93, 222]
    n_visu = sum(p.numel() for p in opt.param_groups[1]["params"]); n_rest = sum(p.numel() for p in opt.param_groups[0]["params"])
    assert n_visu + n_rest == sum(p.numel() for p in m.parameters())
    assert n_visu == 61949149                               # the whole backbone incl. the dead YOLO heads
    assert sum(p.numel() for p in opt.param_groups[1]["params"] if p.requires_grad) == 61949149 - 6687485
    lr = T.adjust_learning_rate(opt, 30, 1e-4, 100, 0.9)
    assert abs(lr - 1e-4 * 0.7 ** 0.9) < 1e-12 and opt.param_groups[1]["lr"] == lr / 10
    assert T.lr_poly(1.0, 0, 10, 0.9) == 1.0


def test_checkpoint_roundtrip_with_module_prefix(tmp_path):
    from dcnet_amd import train as T
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    opt = torch.optim.RMSprop(a.parameters(), lr=0.1)
    a(torch.randn(5, 4)).sum().backward(); opt.step()
    wrapped = {"module." + k: v for k, v in a.state_dict().items()}          # as saved from a DDP wrapper
    path = T.save_checkpoint({"epoch": 7, "state_dict": wrapped, "best_loss": 1.5, "optimizer": opt.state_dict()},
                             True, "t", str(tmp_path))
    assert os.path.exists(os.path.join(str(tmp_path), "t_model_best.pth.tar"))
    b = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    opt_b = torch.optim.RMSprop(b.parameters(), lr=0.1)
    ep, best = T.load_checkpoint(b, path, opt_b)
    assert (ep, best) == (7, 1.5)
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k
    c = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(5))  # shape mismatch on the BN: skipped
    assert T.load_pretrain(c, path) == 3                     # Linear weight + bias + the shape-less num_batches_tracked


def _sha(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def test_darknet_weights_file_pinned_to_the_reference(tmp_path):
    """tests/golden/formats_ref.json (oracle/make_format_goldens.py): the file ``save_weights`` writes from the synthetic
    state_dict was read back, tensor for tensor, by the REFERENCE's ``Darknet.load_weights``; and the reference's own
    ``save_weights`` output equals ``save_weights(reference_layout=True)`` byte for byte.  Here: the same files again."""
    from dcnet_amd.darknet import Darknet
    from dcnet_amd.utils.synth import synth_state_dict
    with open(os.path.join(GOLD, "formats_ref.json")) as f:
        pin = json.load(f)
    sd = synth_state_dict(ref_shapes(256), seed=0)
    a = Darknet(config_path=os.path.join(ROOT, "model", "yolov3.cfg"))
    a.load_state_dict({k[len("visumodel."):]: v for k, v in sd.items() if k.startswith("visumodel.")}, strict=True)
    a.seen = pin["weights_product_file_read_by_reference"]["seen"]
    p1, p2 = str(tmp_path / "all.weights"), str(tmp_path / "ref.weights")
    a.save_weights(p1); a.save_weights(p2, reference_layout=True)
    assert os.path.getsize(p1) == pin["weights_product_file_read_by_reference"]["bytes"] == 248007048
    assert _sha(p1) == pin["weights_product_file_read_by_reference"]["sha256"]
    assert os.path.getsize(p2) == pin["weights_reference_writer_file"]["bytes"]
    assert _sha(p2) == pin["weights_reference_writer_file"]["sha256"]
    b = Darknet(config_path="")
    b.load_weights(p1)
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        if "num_batches_tracked" not in k:
            assert torch.equal(va, vb), k


def test_checkpoint_in_the_reference_layout_loads_with_its_optimizer(tmp_path):
    """A ``.pth.tar`` as the reference writes it (train_DCNet.py:552-557: 'module.'-prefixed keys, torch.optim.RMSprop over
    ALL parameters in two groups) resumes into the product model + fused optimizer after freeze_gradless, and the product's
    optimizer state goes back into an all-parameter torch.optim.RMSprop.  Layout facts (key list digest, group
    sizes, tensor checksums) are those of a checkpoint the real reference wrote (tests/golden/formats_ref.json)."""
    import hashlib
    import zlib
    from dcnet_amd import train as T
    from dcnet_amd.parallel import freeze_gradless
    from dcnet_amd.utils.synth import synth_state_dict
    from model.DCNet_model import grounding_model
    with open(os.path.join(GOLD, "formats_ref.json")) as f:
        pin = json.load(f)["checkpoint_reference_writer"]
    mk = lambda: grounding_model(corpus=list(range(1000)), emb_size=512, img_size=256, weights_path=None,
                                 config_path=os.path.join(ROOT, "model", "yolov3.cfg"))
    src = mk()
    src.load_state_dict(synth_state_dict(ref_shapes(256), seed=0), strict=True)
    wrapped = torch.nn.Sequential(); wrapped.add_module("module", src)
    keys = list(wrapped.state_dict().keys())
    assert len(keys) == pin["n_keys"] and hashlib.sha256("\n".join(keys).encode()).hexdigest() == pin["keys_sha256"]
    for k, c in pin["crc32"].items():
        assert zlib.crc32(wrapped.state_dict()[k].contiguous().numpy().tobytes()) == c, k
    visu = list(src.visumodel.parameters()); ids = {id(p) for p in visu}
    rest = [p for p in wrapped.parameters() if id(p) not in ids]
    ref_opt = torch.optim.RMSprop([{"params": rest}, {"params": visu, "lr": 1e-5}], lr=1e-4, weight_decay=0.0005)
    assert [len(g["params"]) for g in ref_opt.param_groups] == pin["optimizer_group_sizes"] == [93, 222]
    nograd = {"module." + str(k) for k in np.load(os.path.join(GOLD, "train_S256_N4.npz"), allow_pickle=True)["nograd"]}
    g = torch.Generator().manual_seed(5)
    for k, p in wrapped.named_parameters():
        if k not in nograd:
            p.grad = torch.randn(p.shape, generator=g) * 1e-3
    for g_ in ref_opt.param_groups:
        g_["lr"] = 0.0                                      # fill the state, keep the weights (as the fixture's writer did)
    ref_opt.step()
    ref_opt.param_groups[0]["lr"], ref_opt.param_groups[1]["lr"] = 1e-4, 1e-5
    assert len(ref_opt.state_dict()["state"]) == pin["optimizer_state_entries"]
    for i, c in pin["square_avg_crc32"].items():      # the reference's RMSprop wrote exactly these buffers
        assert zlib.crc32(ref_opt.state_dict()["state"][int(i)]["square_avg"].contiguous().numpy().tobytes()) == c, i
    path = T.save_checkpoint({"epoch": 7, "state_dict": wrapped.state_dict(), "best_loss": 0.125, "optimizer": ref_opt.state_dict()},
                             False, "pin", str(tmp_path))
    dst = mk()
    freeze_gradless(dst)
    opt = T.make_optimizer(dst, 1e-4)
    assert T.load_checkpoint(dst, path, opt) == (7, 0.125)
    for k, v in wrapped.state_dict().items():
        assert torch.equal(dst.state_dict()[k[7:]], v), k
    st = opt.state_dict()
    assert [len(g_["params"]) for g_ in st["param_groups"]] == [93, 222] and len(st["state"]) == pin["optimizer_state_entries"]
    # and back: the product's optimizer entry loads into the reference's optimizer class
    back = torch.optim.RMSprop([{"params": rest}, {"params": visu, "lr": 1e-5}], lr=1e-4, weight_decay=0.0005)
    back.load_state_dict(st)
    for i, e in ref_opt.state_dict()["state"].items():
        assert torch.equal(back.state_dict()["state"][i]["square_avg"], e["square_avg"])


def test_oracle_decode_matches_the_reference_fixture():
    """tests/golden/decode_ref.npz holds what the reference's validate_epoch (train_DCNet.py:764-816) decoded."""
    from oracle import dcnet_oracle as O
    g = np.load(os.path.join(GOLD, "decode_ref.npz"))
    for tag in ("256", "416"):
        outbox = [torch.from_numpy(g[f"outbox{s}_{tag}"]) for s in range(3)]
        boxes = O.decode_boxes(outbox, int(tag))
        assert float((boxes - torch.from_numpy(g[f"pred_bbox_{tag}"])).abs().max()) < 1e-4
        iou = O.bbox_iou_xyxy(boxes, torch.from_numpy(g[f"gt_bbox_{tag}"]))
        assert torch.allclose(iou, torch.from_numpy(g[f"iou_{tag}"]), atol=1e-6)
        assert abs(float((iou > 0.5).float().mean()) - float(g[f"accu_{tag}"])) < 1e-6


def test_bench_line_keeps_north_star_numbers_at_benchmark_size():
    """bench.compact_line on a canned result of the benchmark geometry (the strings and digit counts of a real driver run):
    the line stays under 2 KB AND carries, inside `roofline`, what north_star asks for — the dominant kernel alone and in the
    step, the conv engine against 838.9, the cross-modal scoring against 8 TB/s, the exact-arithmetic alternatives, the clock."""
    import json
    import bench
    res = {"metric": "clips/sec (T=8, 416x416, bs8) fwd+bwd", "value": 80.7918522545519, "unit": "clips/s", "n_gpus": 1, "steps": 20, "warmup": 5,
           "ms_per_step": 99.01988600031473, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "T=8 416x416 bs8/GPU L=20 fp32, 64 img/GPU/step as pairs, fwd+5 losses+bwd+RMSprop",
                      "arith": "f16x2-split MFMA, fp32 accumulate", "parallelism": "dp1", "ranks_seen": 1, "reducer": "none",
                      "step": "hipGraph replay (fwd+losses+bwd+RMSprop)"},
           "host_queue_ms_per_step": 89.53, "mem_gb": 30.1, "loss": 123.4567, "loss_hex": "0x1.1d65820000000p+4", "src": "9246e1312c10", "host_ms_per_step": {"launch": 5.01, "sampler_thread": 18.2, "gpu_wait": 84.42},
           "roofline": {"bound": "mfma", "kernel": bench.FAMILY[28], "rocprof_match": bench.RP_MATCH[28], "achieved": 334.0, "peak": 838.9,
                        "unit": "TFLOP/s", "frac": 0.3982, "traffic": 389711014.40000004, "traffic_ratio": 1.933, "avg_launch_ms": 0.382,
                        "ms_per_step": 22.92, "launches_per_step": 60.0, "alg_bytes_per_launch": 201624781, "hbm_frac_algorithmic": 0.066,
                        "binding_frac": 0.3982, "in_step": {"avg_launch_ms": 0.4176, "frac": 0.3642, "shared": 0.411},
                        "bf16s_clips_s": 138.52, "bf16s_ms_per_step": 57.75, "bf16s_criterion": "15/16"},
           "flop_dominant": {"kernel": bench.FAMILY[28], "frac": 0.3982, "ms_per_step": 22.92, "binding_frac": 0.3982},
           "hbm_scoring": {"kernel": bench.NAMES[8], "achieved": 5188.3, "peak": 8000.0, "unit": "GB/s", "frac": 0.6485},
           "conv_engine": {"tflop_per_step": 18.87, "tflops_over_kernel_time": 258.9, "frac_of_838.9": 0.3086, "frac_of_fp32_mfma_157.3": 1.646,
                           "tflops_over_step_wall": 190.6},
           "bn_passes_ms_per_step": 13.71,
           "alt": {"exclusive_ms": 112.0, "bf16x3_ms": 152.6, "native_fp32_ms": 221.8, "bf16_ms": 102.7, "bf16s_ms": 67.9, "fp8_ms": 109.1},
           "cpu_baseline": {"value": 0.2383, "unit": "clips/s", "cores": 8, "kind": "port",
                            "sample": "oracle port fwd+5 losses+bwd, 1 clip T=8 416x416, median of 3 steps, 128 cpus on host",
                            "gpu_vs_oracle_max_abs_err": 0.00019, "acc_at_0.5_vs_oracle_boxes": 1.0},
           "full": "profiles/bench_full_latest.json", "sclk_mhz": 2104}
    line = bench.compact_line(res)
    assert len(line) < 2000, len(line)
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "host_ms_per_step"):
        assert k in out, k
    rf = out["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "in_step_frac", "in_step_ms", "conv_engine_frac",
              "hbm_scoring_frac", "hbm_scoring_gbs", "native_fp32_ms", "bf16x3_ms", "bf16_ms", "fp8_ms", "alone_ms", "sclk_mhz",
              "step_tflops", "step_frac", "bf16s_clips_s", "bf16s_ms_per_step", "bf16s_criterion"):
        assert k in rf, k
    assert "rocprof_match" not in rf and out["src"] == "9246e1312c10"      # (the match strings live in the full JSON; the line names its sources)
    assert rf["in_step_frac"] == 0.3642 and rf["hbm_scoring_frac"] == 0.6485 and rf["conv_engine_frac"] == 0.3086 and rf["sclk_mhz"] == 2104
    assert rf["step_tflops"] == 190.6 and abs(rf["step_frac"] - 190.6 / 838.9) < 1e-3        # the replayed step's FLOP over its wall time
    assert all(len(v) <= 120 for v in (out["config"]["workload"], rf["kernel"], rf["note"], out["cpu_baseline"]["sample"]))   # the driver cuts strings
    # an absurdly long line sheds bookkeeping keys, never the contract or the north_star numbers
    res["config"]["workload"] = "x" * 700
    out = json.loads(bench.compact_line(res))
    assert "loss_hex" not in out and out["roofline"]["hbm_scoring_frac"] == 0.6485 and "cpu_baseline" in out
    assert out["roofline"]["bf16s_ms_per_step"] == 57.75


def test_graph_queue_model_on_a_small_dag(tmp_path):
    """tools/graph_sched.py on a hand-written dump in the runtime's format: depth first from the root, the first dependent of a node keeps
    its queue, the k-th gets queue + k (mod 4), a node reached twice keeps its first queue — the rule the captured step's stream
    schedule is built on (the GPU test checks it against the runtime's own dump of the real step)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("graph_sched", os.path.join(root, "tools", "graph_sched.py"))
    gs = importlib.util.module_from_spec(spec); spec.loader.exec_module(gs)

    def node(i, name, sid):
        return f'"graph_1_node_{i}"[style="bold"shape="octagon"label="{i}\n{name}\nStreamId:{sid}\nSignalIsRequired: false\nDeviceId:0"];\n'

    # 0 -> 1 -> 2 -> 5 (main chain), 1 -> 3 -> 4 -> 5 (a side branch captured behind the main continuation), 2 -> 6 and 2 -> 7 (two more forks)
    edges = [(0, 1), (1, 2), (1, 3), (2, 5), (2, 6), (2, 7), (3, 4), (4, 5), (6, 5), (7, 5)]
    want = {0: 0, 1: 0, 2: 0, 5: 0, 3: 1, 4: 1, 6: 1, 7: 2}
    text = "digraph dot {\nsubgraph cluster_1 {\n" + "".join(node(i, f"k{i}", want[i]) for i in range(8))
    text += "".join(f'"graph_1_node_{a}" -> "graph_1_node_{b}";\n' for a, b in edges) + "}\n}\n"
    path = tmp_path / "small.dot"
    path.write_text(text)
    nodes, order, ed = gs.parse(str(path))
    assert len(nodes) == 8 and len(ed) == len(edges)
    sim = gs.simulate(nodes, order, ed)
    assert {nodes[n]["idx"]: s for n, s in sim.items()} == want
    # the same DAG with the side branch captured FIRST: it inherits the main chain's queue and the main chain moves on
    edges2 = [(0, 1), (1, 3), (1, 2)] + [e for e in edges if e not in ((0, 1), (1, 2), (1, 3))]
    text2 = text.split('"graph_1_node_0" ->')[0] + "".join(f'"graph_1_node_{a}" -> "graph_1_node_{b}";\n' for a, b in edges2) + "}\n}\n"
    path.write_text(text2)
    nodes, order, ed = gs.parse(str(path))
    sim2 = {nodes[n]["idx"]: s for n, s in gs.simulate(nodes, order, ed).items()}
    assert sim2[3] == 0 and sim2[4] == 0 and sim2[5] == 0 and sim2[2] == 1


def test_profiling_tags_are_named_and_not_shared_between_kernel_files():
    """Every literal tag a csrc file hands to prof_begin() has a name in bench.py, fits DCN_PROF_TAGS, and belongs to ONE file — two kernels
    of different files on one tag add their times and work up under the first one's name (round 5: conv2b and the bf16 strip kernel both
    used 47).  Tag 20 (bf16-operand weight gradients of wgrad.hip and wgrad3.hip) is the one deliberate exception."""
    import glob
    import os
    import re
    import bench
    csrc = os.path.join(ROOT, "dcnet_amd", "csrc")
    owners = {}
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        src = open(path).read()
        for m in re.finditer(r"prof_begin\(([^;]*?),\s*[^,;]*,\s*\(?\s*(?:hipStream_t\)\s*)?stream", src):
            expr = re.sub(r"(==|<=|>=|<|>)\s*\d+", "", m.group(1))          # (literals of conditions: NP == 1, p.M < 1024, ...)
            for lit in re.findall(r"(?<![\w.])(\d+)(?![\w.])", expr):
                owners.setdefault(int(lit), set()).add(os.path.basename(path))
    assert owners, "no prof_begin() call found: the pattern of this test is stale"
    tags = set(owners)
    for t in sorted(tags):
        assert t < bench.NT, f"tag {t} does not fit DCN_PROF_TAGS = {bench.NT}"
        assert t in bench.NAMES, f"tag {t} ({sorted(owners[t])}) has no name in bench.NAMES"
        if t != 20:
            assert len(owners[t]) == 1, f"tag {t} is used by {sorted(owners[t])}"
    with open(os.path.join(csrc, "prof.h")) as f:
        assert int(re.search(r"#define\s+DCN_PROF_TAGS\s+(\d+)", f.read()).group(1)) == bench.NT


def test_build_cache_and_source_hash_are_keyed_on_the_compile_flags(monkeypatch, tmp_path):
    """Round-5 advice: an experiment build (DCN_EXTRA_FLAGS=-DC3_ABL=1: results wrong by construction) must neither pass as an up-to-date
    default build nor stamp a bench line with the default build's source hash.  dcnet_amd/build.py keys its objects on the flags
    (build/FLAGS.stamp: other flags -> everything is stale) and utils/srchash.py folds the stamp of the flags the library WAS built with."""
    import importlib
    import dcnet_amd.build as b
    import dcnet_amd.utils.srchash as sh
    k0 = b.flags_key()
    monkeypatch.setenv("DCN_EXTRA_FLAGS", "-DC3_ABL=1")
    try:
        assert importlib.reload(b).flags_key() != k0
    finally:
        monkeypatch.delenv("DCN_EXTRA_FLAGS")
        assert importlib.reload(b).flags_key() == k0
    # the hash follows the stamp file, not the environment: point the module at a copy of the package tree's stamp
    h0 = sh.kernel_sources_hash()
    stamp = os.path.join(os.path.dirname(b.__file__), "build", "FLAGS.stamp")
    if os.path.exists(stamp):                      # (a built tree: the usual case; build() writes it)
        assert b.built_flags_key() == k0, "the library in the tree was not built with the default flags"
        old = open(stamp).read()
        try:
            with open(stamp, "w") as f:
                f.write("0123456789ab\n")
            assert sh.kernel_sources_hash() != h0
        finally:
            with open(stamp, "w") as f:
                f.write(old)
        assert sh.kernel_sources_hash() == h0
