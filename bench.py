"""DCNet hot-path benchmark: clips/s, forward + backward (+ optimizer step), T=8 frames of
416x416, batch 8 clips per GPU (BASELINE.json configs[1]: fp32, synthetic data, random init).

    python bench.py --gpus 1 --steps 5 --warmup 2        (always through the interpreter — under rocprofv3: `-- python3 bench.py`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; clips are sharded data-parallel (weak scaling: 8 clips per GPU), gradients are
all-reduced by RCCL through DistributedDataParallel.  A step = forward of the drop-in
``grounding_model`` on 64 images (= 8 clips x T 8, pair semantics: SURVEY.md F1), the five training
losses, backward, RMSprop step — nothing skipped.  Prints ONE JSON line on rank 0.

The ``roofline`` object is measured live: the library records a HIP event pair around every launch
of the conv engine on the launch stream during the timed steps (dcn_prof_*).  ``cpu_baseline`` times
the CPU oracle (oracle/, a restatement pinned against the reference) on 1 clip of the same shape.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, == fp32 vector rate
PEAK_BF16_MFMA_TFLOPS = 2516.6     # 256 CU x 4 SIMD x 1024 FLOP/clk (v_mfma_f32_32x32x16_bf16, 32 cycles) x 2.4 GHz
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6      # bf16 three-piece split: six bf16 MFMAs per fp32 product term
PEAK_H2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3         # f16 two-piece split (the default): three f16 MFMAs (same rate) per product
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=8, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=8, help="T")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exclusive-steps", type=int, default=2,
                    help="untimed steps after the timed region with all side streams off, for the per-kernel "
                         "exclusive duration (0 = skip, e.g. under rocprofv3 so its averages match the timed region)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DDP/RCCL even at world size 1 (exercises the hooks)")
    ap.add_argument("--tune", type=str, default="", help="key=value[,key=value]: dcn_set_tuning knobs applied before the run (experiments)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: every rank uses cuda:0 and the process group runs on gloo (RCCL refuses "
                         "two ranks per device) — exercises broadcast, sharded seeds, the reducer and the max-over-ranks timing")
    ap.add_argument("--reducer", choices=["overlap", "ddp"], default="overlap",
                    help="multi-GPU gradient averaging: 'overlap' = dcnet_amd.parallel.OverlappedGradReducer (buckets all-reduced on a "
                         "communication stream while the backbone's backward is still running), 'ddp' = torch DDP wrapper")
    return ap.parse_args()


def parity_check(size: int = 256, n: int = 4):
    """The metric's second half (Acc@0.5 IoU), as agreement with the oracle: decoded boxes of the HIP model vs the CPU
    oracle on a small seeded eval batch (the oracle is the checker here, as in smoke()); part of the cpu_baseline leg."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import build_product, maxdiff, synth_sd
    from dcnet_amd.utils.synth import synth_inputs
    from oracle import dcnet_oracle as O
    dev = torch.device("cuda", torch.cuda.current_device())
    sd = synth_sd(size)
    image, word_id, word_mask = synth_inputs(n, size, seed=77)
    m = build_product(size, sd, dev).eval()
    with torch.no_grad():
        outbox = m(image.to(dev), word_id.to(dev), word_mask.to(dev))[0]
        o = O.grounding_forward_pairs({k: v.clone() for k, v in sd.items()}, image, word_id, training=False, sample=False)
    boxes = O.decode_boxes([x.cpu() for x in outbox], size)
    ref = O.decode_boxes(o["outbox"], size)
    iou = O.bbox_iou_xyxy(boxes, ref)
    return {"images": n, "size": size, "acc_at_iou_0.5_vs_oracle_boxes": float((iou > 0.5).float().mean()),
            "min_iou_vs_oracle_boxes": float(iou.min()),
            "max_abs_err_outbox": max(maxdiff(a, b) for a, b in zip(outbox, o["outbox"]))}


def cpu_baseline(size: int, frames: int, steps: int):
    """Oracle forward+backward on the host cores, 1 clip (bs 8 would need ~80 GB of host RAM).  The thread count is
    chosen by a short sweep (an eval forward of one frame pair at each of 8/16/32/64 threads): more threads than that only
    slow torch's CPU kernels down on this workload (128 threads measured half the 8-thread rate in round 1)."""
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs, synth_state_dict
    from oracle import dcnet_oracle as O
    from oracle import train_oracle as TO
    import random
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys_256.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    P = sum((size // 32 * 2 ** i) ** 2 for i in range(3))
    shapes["loc_text_embedding.0.weight"] = (512, P)
    sd0 = synth_state_dict(shapes, seed=0)
    image, word_id, _ = synth_inputs(frames, size, seed=1)
    bbox = synth_boxes(frames, size, seed=1)
    had = torch.get_num_threads()
    sweep = {}
    for nt in (8, 16, 32, 64):
        if nt > (os.cpu_count() or 8):
            continue
        torch.set_num_threads(nt)
        ts = []
        for it in range(3):
            t0 = time.perf_counter()
            with torch.no_grad():
                O.grounding_forward_pairs({k: v.clone() for k, v in sd0.items()}, image[:2], word_id[:2], training=False, sample=False)
            ts.append(time.perf_counter() - t0)
        sweep[nt] = min(ts[1:])
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = []
    for it in range(steps + 1):
        sd = {k: v.clone() for k, v in sd0.items()}
        for k, v in sd.items():
            if v.dtype == torch.float32 and "running" not in k:
                v.requires_grad_(True)
        random.seed(13)
        t0 = time.perf_counter()
        o = O.grounding_forward_pairs(sd, image, word_id, training=True)
        loss, _ = TO.total_loss(o, bbox, size)
        loss.backward()
        times.append(time.perf_counter() - t0)
    torch.set_num_threads(had)
    t = float(np.median(times[1:]))
    return {"value": 1.0 / t, "unit": "clips/s", "cores": best, "kind": "port",
            "thread_sweep_eval_pair_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "parity": parity_check(),
            "sample": f"oracle/ (CPU restatement pinned to the reference) fwd+5 losses+bwd on 1 clip T={frames} "
                      f"{size}x{size}, {steps} timed steps after 1 warm-up, median, at the best of 8/16/32/64 threads "
                      f"(picked on an eval forward of one frame pair); host has {os.cpu_count()} cpus"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_ddp = world > 1 or args.force_ddp
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"       # keep RCCL's version banner out of stdout (one JSON line only)
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")     # RCCL prints its warnings on stdout by default
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.rehearse:
            torch.distributed.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    from dcnet_amd import losses
    from dcnet_amd.lib import lib
    from dcnet_amd.model import grounding_model
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs

    for kv in [t for t in args.tune.split(",") if t]:
        k_, v_ = kv.split("=")
        lib().set_tuning(k_.encode(), int(v_))
    torch.manual_seed(1234)            # identical initial weights on every rank (DDP also broadcasts rank 0's)
    model = grounding_model(corpus=list(range(1000)), light=False, emb_size=512, coordmap=True,
                            bert_model="bert-base-uncased", dataset="vid", img_size=args.size,
                            config_path=os.path.join(ROOT, "model", "yolov3.cfg"), weights_path=None).to(dev)
    model.train()
    # parameters that never receive a gradient in the reference either (dead YOLO heads F7, feature_map F8):
    # freezing them gives DDP a static graph instead of find_unused_parameters=True (train_DCNet.py:483)
    from dcnet_amd.parallel import freeze_gradless, wrap_ddp
    freeze_gradless(model)
    red = None
    if use_ddp and args.reducer == "overlap":
        from dcnet_amd.parallel import attach_overlapped_reducer
        red = attach_overlapped_reducer(model)
        net = model
    else:
        net = wrap_ddp(model, local_rank) if use_ddp else model
    from dcnet_amd.train import make_optimizer     # the reference's two RMSprop groups (train_DCNet.py:519-534), fused HIP step
    opt = make_optimizer(model, 1e-4)

    n_img = args.clips * args.frames
    image, word_id, word_mask = synth_inputs(n_img, args.size, seed=100 + rank)
    bbox = synth_boxes(n_img, args.size, seed=100 + rank)
    image, word_id, word_mask, bbox = image.to(dev), word_id.to(dev), word_mask.to(dev), bbox.to(dev)
    import random
    random.seed(13 + rank)

    def step():
        out = net(image, word_id, word_mask)
        loss, _ = losses.total_loss(out, bbox, args.size)
        opt.zero_grad(set_to_none=True)
        if red is not None:
            red.begin_step()
        loss.backward()
        if red is not None:
            red.finish()             # heads / language gradients in one flat bucket; joins the communication stream
        opt.step()
        return loss

    def barrier():
        if use_ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    L = lib()
    L.prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    host_dt = time.perf_counter() - t0          # the host has queued every step (no synchronisation inside a step)
    barrier()
    dt = time.perf_counter() - t0
    L.prof_enable(0)
    from dcnet_amd import ops as _ops_chk
    _ops_chk.check_bilstm(dev)                  # the persistent BiLSTM's sticky error word: a timed-out hand-off must not pass as a result
    if use_ddp:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    NT = 40            # DCN_PROF_TAGS
    alg_bytes = []

    def collect():
        c = (ctypes.c_int64 * NT)(); m = (ctypes.c_double * NT)(); w = (ctypes.c_double * NT)(); by = (ctypes.c_double * NT)()
        L.prof_collect(ctypes.addressof(c), ctypes.addressof(m), ctypes.addressof(w), ctypes.addressof(by))
        alg_bytes.clear(); alg_bytes.extend(list(by))
        return list(c), list(m), list(w)

    counts, ms, work = collect()
    timed_bytes = list(alg_bytes)

    # Untimed passes after the timed region (every rank runs them, so DDP stays in step):
    #  * "exclusive": the product configuration with the weight-gradient stream switched off.  The timed region
    #    overlaps weight-gradient GEMMs with the data-gradient chain, so a launch's event-to-event time there
    #    includes the time it shared the CUs; alone on the GPU the duration speaks about the kernel itself.
    #  * "native_fp32": the same step with dcn_set_tuning("precision", 0) — every tile on v_mfma_f32_32x32x2_f32.
    from dcnet_amd import ops as _ops

    def extra_pass(precision: int):
        was = (_ops.WGRAD_SIDE, model.language_stream, model.sampling_stream)
        _ops.WGRAD_SIDE = False; model.language_stream = False; model.sampling_stream = False
        _ops.set_precision({v: k for k, v in _ops.PRECISIONS.items()}[precision])       # (the host side follows the mode: which
        step(); barrier()                                                               #  operand banks / abs-max words it prepares)
        L.prof_enable(1)
        t1 = time.perf_counter()
        for _ in range(args.exclusive_steps):
            step()
        barrier()
        el = time.perf_counter() - t1
        L.prof_enable(0)
        _ops.WGRAD_SIDE, model.language_stream, model.sampling_stream = was
        _ops.set_precision("fp32")
        return collect() + (el / args.exclusive_steps * 1e3,)

    excl = native = bf16 = fp8 = bf16x3 = None
    if args.exclusive_steps > 0:
        excl = extra_pass(4)
        bf16x3 = extra_pass(1)
        native = extra_pass(0)
        bf16 = extra_pass(2)
        fp8 = extra_pass(3)

    if rank == 0:
        clips_total = args.clips * world * args.steps
        names = {0: "igemm_kernel<128,128,2,2,0,false,16>", 15: "igemm_kernel<128,128,2,2,0,false,32>",
                 1: "igemm_kernel<128,64,2,2,0>", 2: "igemm_kernel<256,32,4,1,0>",
                 3: "igemm_kernel<128,128,2,2,1>", 4: "igemm_kernel<128,64,2,2,1>", 5: "wgrad_kernel<*>",
                 6: "igemm_kernel<64,128,2,2,0>", 7: "igemm_kernel<64,128,2,2,1>",
                 8: "l2norm_score_fwd_kernel", 9: "l2norm_score_bwd_kernel", 10: "scale_act_kernel",
                 11: "bn_act_bwd_apply_kernel", 12: "exp_sums_kernel",
                 13: "igemm_kernel<...> (LSTM-step GEMMs, <1024 rows, side stream)",
                 14: "wgrad_kernel<...> (LSTM-step GEMMs, side stream)",
                 16: "igemm_kernel<128,128,2,2,0,false,16,true>",
                 17: "wgrad_kernel<128,128,16,true>",
                 18: "igemm_kernel<256,64,4,1,0,false,16,true>",
                 19: "igemm_kernel<*,*,*,*,0,false,32,true,0,1> (bf16 operands)",
                 20: "wgrad_kernel<128,128,16,true,0,1> (bf16 operands)",
                 21: "igemm_kernel<128,128,2,2,1,false,16,true> (NN)",
                 22: "channel_partials_kernel<1>",
                 23: "igemm_kernel<*,*,*,*,0,false,32,true,0,1,1,true> (fp8 operands)",
                 # 24-27: the f16 two-piece split (fp32 accuracy, three MFMAs per product): the default arithmetic
                 24: "igemm_kernel<128,128,2,2,0,false,16,true,0,2>",
                 25: "wgrad_kernel<128,128,16,true,0,2>",
                 26: "igemm_kernel<256,64,4,1,0,false,16,true,0,2>",
                 27: "igemm_kernel<128,128,2,2,1,false,16,true,0,2>",
                 # 28: the 3x3 stride-1 layers (forward and data gradient): f16 split with the activation strip resident in LDS
                 28: "conv3_kernel<4,2,4>", 29: "conv3_kernel<2,2,4>",
                 30: "reduce_slabs_kernel", 31: "dA_kernel",
                 # 32: weight gradient of the 3x3 stride-1 layers, one filter row per workgroup (f16 split)
                 32: "wgrad3_kernel",
                 # 33: the strip kernel with bf16 operands (one plane, one MFMA per product): the bf16-operand mode's 3x3 layers
                 33: "conv3_kernel<*,2,4,1> (bf16 operands)",
                 # 34: the stem (4-channel image -> 32 filters) directly on the vector ALU, HBM-priced
                 34: "stem_kernel"}
        flop_tags = set(range(8)) | {13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24, 25, 26, 27, 28, 29, 32, 33}
        peak_of = {t: (PEAK_SPLIT_TFLOPS if t in (16, 17, 18, 21) else PEAK_H2_TFLOPS if t in (24, 25, 26, 27, 28, 29, 32)
                       else PEAK_BF16_MFMA_TFLOPS if t in (19, 20, 23, 33) else PEAK_FP32_MFMA_TFLOPS) for t in flop_tags}

        def table(c, m, w, nsteps):
            out = {}
            for t, nm in names.items():
                if c[t]:
                    fl = t in flop_tags
                    rate = w[t] / (m[t] * 1e-3) / (1e12 if fl else 1e9)
                    out[nm] = {"launches_per_step": c[t] / nsteps, "avg_ms": m[t] / c[t], "ms_per_step": m[t] / nsteps,
                               "achieved": rate, "unit": "TFLOP/s" if fl else "GB/s",
                               "frac": rate / (peak_of[t] if fl else PEAK_HBM_GBS)}
            return out

        kern = table(counts, ms, work, args.steps)
        mm_tags = sorted(flop_tags - {13, 14})
        # the kernel that carries most of the step's FLOPs; the two tile builds of the strip kernel count as one family
        fam = lambda t: work[28] + work[29] if t in (28, 29) else work[t]
        dom = max(mm_tags, key=lambda t: (fam(t), work[t]))
        traffic = None; traffic_src = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_file):     # HBM bytes per launch from the last rocprofv3 --pmc passes (not measurable live)
            with open(pmc_file) as f:
                pmc = json.load(f)
            if pmc.get("kernel") == names[dom]:
                traffic = pmc.get("hbm_bytes_per_launch")
                traffic_src = "profiles/pmc_latest.json (rocprofv3 --pmc passes of an earlier run of this command, not this run)"
        alg_b = timed_bytes[dom] / counts[dom] if counts[dom] else None
        mfma_ms = sum(ms[t] for t in mm_tags); mfma_work = sum(work[t] for t in mm_tags)
        ach = work[dom] / (ms[dom] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": names[dom], "achieved": ach, "peak": peak_of[dom], "unit": "TFLOP/s",
                    "frac": ach / peak_of[dom], "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": alg_b, "traffic_ratio": (traffic / alg_b) if (traffic and alg_b) else None,
                    "avg_launch_ms": ms[dom] / counts[dom], "flop_per_launch": work[dom] / counts[dom],
                    "peak_note": ("fp32 operands as 2 f16 pieces (per-tensor power-of-two scale), 3 cross terms on "
                                  "v_mfma_f32_32x32x16_f16: peak = 2516.6 TFLOP/s dense f16 / 3 = 838.9 algorithmic fp32 TFLOP/s")
                                 if dom in (24, 25, 26, 27, 28, 29, 32) else
                                 ("fp32 operands as 3 exact bf16 pieces, 6 cross terms on v_mfma_f32_32x32x16_bf16: peak = "
                                  "2516.6 TFLOP/s dense bf16 / 6 = 419.4 algorithmic fp32 TFLOP/s (fp32 pipe: 157.3)")
                                 if dom in (16, 17, 18, 21) else "v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s",
                    "frac_of_bf16x3_ceiling": ach / PEAK_SPLIT_TFLOPS, "frac_of_fp32_mfma_peak": ach / PEAK_FP32_MFMA_TFLOPS,
                    "overlap": "the timed region runs the weight-gradient GEMMs on a second stream beside the data-gradient "
                               "chain; per-launch durations here include the time a launch shared the CUs",
                    # every FLOP the MFMA kernels were asked for in a step over the whole step's wall time
                    "step_mfma": {"tflop_per_step": mfma_work / args.steps / 1e12, "achieved": mfma_work / 1e12 / dt,
                                  "frac_of_fp32_pipe": mfma_work / 1e12 / dt / PEAK_FP32_MFMA_TFLOPS},
                    "hbm_scoring": {"kernel": names[8], "achieved": kern.get(names[8], {}).get("achieved"),
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": (kern[names[8]]["achieved"] / PEAK_HBM_GBS) if names[8] in kern else None},
                    "kernels": kern}
        if excl is not None:
            c2, m2, w2, ms_step = excl
            if c2[dom]:
                a2 = w2[dom] / (m2[dom] * 1e-3) / 1e12
                roofline["exclusive"] = {"kernel": names[dom], "achieved": a2, "peak": peak_of[dom], "frac": a2 / peak_of[dom],
                                         "avg_launch_ms": m2[dom] / c2[dom], "steps": args.exclusive_steps, "ms_per_step": ms_step,
                                         "note": "same launches with the weight-gradient stream switched off (untimed pass)",
                                         "kernels": table(c2, m2, w2, args.exclusive_steps)}
            cb, mb, wb, ms_stepb = bf16x3
            db = max(mm_tags, key=lambda t: wb[t])
            ab = wb[db] / (mb[db] * 1e-3) / 1e12
            roofline["fp32_bf16x3"] = {"kernel": names[db], "achieved": ab, "peak": PEAK_SPLIT_TFLOPS, "frac": ab / PEAK_SPLIT_TFLOPS,
                                       "avg_launch_ms": mb[db] / cb[db], "ms_per_step": ms_stepb,
                                       "clips_per_s": args.clips * world / (ms_stepb * 1e-3),
                                       "note": "dcn_set_tuning('precision', 1): the round-1 arithmetic (three bf16 pieces, six MFMAs per "
                                               "product), side streams off (untimed pass)",
                                       "kernels": table(cb, mb, wb, args.exclusive_steps)}
            c3, m3, w3, ms_step3 = native
            d3 = max(mm_tags, key=lambda t: w3[t])
            a3 = w3[d3] / (m3[d3] * 1e-3) / 1e12
            roofline["native_fp32"] = {"kernel": names[d3], "achieved": a3, "peak": PEAK_FP32_MFMA_TFLOPS,
                                       "frac": a3 / PEAK_FP32_MFMA_TFLOPS, "avg_launch_ms": m3[d3] / c3[d3],
                                       "ms_per_step": ms_step3, "clips_per_s": args.clips * world / (ms_step3 * 1e-3),
                                       "note": "dcn_set_tuning('precision', 0): every tile on v_mfma_f32_32x32x2_f32, "
                                               "weight-gradient stream off (untimed pass)",
                                       "kernels": table(c3, m3, w3, args.exclusive_steps)}
            c4, m4, w4, ms_step4 = bf16
            d4 = max(mm_tags, key=lambda t: w4[t])
            a4 = w4[d4] / (m4[d4] * 1e-3) / 1e12
            roofline["bf16_operands"] = {"kernel": names[d4], "achieved": a4, "peak": PEAK_BF16_MFMA_TFLOPS,
                                         "frac": a4 / PEAK_BF16_MFMA_TFLOPS, "avg_launch_ms": m4[d4] / c4[d4],
                                         "ms_per_step": ms_step4, "clips_per_s": args.clips * world / (ms_step4 * 1e-3),
                                         "note": "dcn_set_tuning('precision', 2) = BASELINE.json configs[2] on one GPU: bf16 operands "
                                                 "(RNE), one MFMA per product, fp32 accumulate, fp32 tensors; REDUCED precision, "
                                                 "reported beside the fp32 headline, never as `value` (untimed pass, weight-gradient "
                                                 "stream off)",
                                         "kernels": table(c4, m4, w4, args.exclusive_steps)}
            c5, m5, w5, ms_step5 = fp8
            d5 = max(mm_tags, key=lambda t: w5[t])
            a5 = w5[d5] / (m5[d5] * 1e-3) / 1e12
            roofline["fp8_operands"] = {"kernel": names[d5], "achieved": a5, "peak": PEAK_BF16_MFMA_TFLOPS,
                                        "frac": a5 / PEAK_BF16_MFMA_TFLOPS, "avg_launch_ms": m5[d5] / c5[d5],
                                        "ms_per_step": ms_step5, "clips_per_s": args.clips * world / (ms_step5 * 1e-3),
                                        "note": "ops.set_precision('fp8') = BASELINE.json configs[4] at batch 8: fp8 e4m3 operands with "
                                                "per-tensor power-of-two scales (non-scaled fp8 MFMA: bf16 rate) in forward and data "
                                                "gradient, bf16 operands in the weight gradient, fp32 accumulate and tensors; REDUCED "
                                                "precision, reported beside the fp32 headline, never as `value` (untimed pass)",
                                        "kernels": table(c5, m5, w5, args.exclusive_steps)}
        res = {"metric": f"clips/sec (T={args.frames}, {args.size}x{args.size}, bs{args.clips}) fwd+bwd", "value": clips_total / dt, "unit": "clips/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "precision": "fp32 tensors and fp32 accumulation everywhere; the wide GEMM tiles multiply on the f16 matrix pipe with "
                            "each operand scaled by a per-tensor power of two and cut into 2 f16 pieces (11 + 11 bits; l*h + h*l + h*h): "
                            "error vs fp64 <= that of the fp32 MFMA instruction (tests/test_ops_gpu.py::test_split_pipe_is_fp32_accurate)",
               "config": {"workload": f"T={args.frames} {args.size}x{args.size} bs{args.clips} clips/GPU, 20-token query, fp32, "
                                      f"pair semantics ({n_img} images/GPU/step), fwd + 5 losses + bwd + RMSprop",
                          "images_per_gpu": n_img, "parallelism": f"dp{world}"},
               "host_queue_ms_per_step": host_dt / args.steps * 1e3,
               "memory": {"max_allocated_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
                          "reserved_gb": torch.cuda.memory_reserved(dev) / 2 ** 30,
                          "alloc_retries": torch.cuda.memory_stats(dev).get("num_alloc_retries", 0)},
               "loss": float(last.detach()), "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.size, args.frames, args.cpu_steps)
        sys.stdout.write(json.dumps(res) + "\n")
        sys.stdout.flush()
    if use_ddp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
