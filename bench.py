"""DCNet hot-path benchmark: clips/s, forward + backward (+ optimizer step), T=8 frames of
416x416, batch 8 clips per GPU (BASELINE.json configs[1]: fp32, synthetic data, random init).

    python bench.py --gpus 1 --steps 5 --warmup 2        (always through the interpreter — under rocprofv3: `-- python3 bench.py`)
    python bench.py --gpus N ...                         (no launcher in the environment: spawns N ranks through
                                                          `python -m torch.distributed.run`, before this process touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; clips are sharded data-parallel (weak scaling: 8 clips per GPU), gradients are averaged over RCCL.
A step = forward of the drop-in ``grounding_model`` on 64 images (= 8 clips x T 8, pair semantics: SURVEY.md F1), the five
training losses, backward, RMSprop step — nothing skipped.  The timed region runs the step as ONE replayed hipGraph
(dcnet_amd.graph.GraphedTrainStep; ``--graph off`` = the eager Python step) and contains no profiling.

Rank 0 prints ONE compact JSON line (< 2 KB).  The per-kernel tables behind it — every conv-engine launch of an untimed
profiled pass with a HIP event pair on its launch stream (dcn_prof_*) — go to ``profiles/bench_full_latest.json`` (and
``gpurun_out/`` when present).  ``cpu_baseline`` times the CPU oracle (oracle/, a restatement pinned against the reference)
on 1 clip of the same shape.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
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

# dcn_prof tags (csrc/prof.h) -> kernel names as rocprofv3 prints them (template arguments abbreviated)
NAMES = {0: "igemm_kernel<128,128,2,2,0,false,16>", 15: "igemm_kernel<128,128,2,2,0,false,32>",
         1: "igemm_kernel<128,64,2,2,0>", 2: "igemm_kernel<256,32,4,1,0>",
         3: "igemm_kernel<128,128,2,2,1>", 4: "igemm_kernel<128,64,2,2,1>", 5: "wgrad_kernel<*> (fp32 pipe, narrow tiles)",
         6: "igemm_kernel<64,128,2,2,0>", 7: "igemm_kernel<64,128,2,2,1>",
         8: "l2norm_score_fwd_kernel", 9: "l2norm_score_bwd_kernel", 10: "scale_act_kernel",
         11: "bn_act_bwd_apply_kernel", 12: "exp_sums_kernel",
         13: "igemm_kernel<...> (language-branch GEMMs, <1024 rows, side stream)",
         14: "wgrad_kernel<...> (language-branch GEMMs, side stream)",
         16: "igemm_kernel<128,128,2,2,0,false,16,true> (bf16x3)", 17: "wgrad_kernel<128,128,16,true> (bf16x3)",
         18: "igemm_kernel<256,64,4,1,0,false,16,true> (bf16x3)",
         19: "igemm_kernel<*,*,*,*,0,false,32,true,0,1> (bf16 operands)", 20: "wgrad_kernel<128,128,16,true,0,1> (bf16 operands)",
         21: "igemm_kernel<128,128,2,2,1,false,16,true> (NN, bf16x3)", 22: "channel_partials_kernel<1>",
         23: "igemm_kernel<*,*,*,*,0,false,32,true,0,1,1,true> (fp8 operands)",
         24: "igemm_kernel<128,128,2,2,0,false,16,true,0,2> (1x1 / stride-2 / co-attention NT tiles, f16 split)",
         25: "wgrad_kernel<128,128,16,true,0,2>", 26: "igemm_kernel<256,64|32,4,1,0,false,16,true,0,2>",
         27: "igemm_kernel<128,128,2,2,1,false,16,true,0,2> (NN)",
         28: "conv3_kernel<4,2,4>", 29: "conv3_kernel<2,2,4>", 30: "reduce_slabs_kernel", 31: "dA_kernel", 32: "wgrad3_kernel",
         33: "conv3_kernel<*,2,4,1> (bf16 operands)", 34: "stem_kernel", 35: "conv1_kernel (1x1, f16 split)",
         36: "wgrad9_kernel (32 -> 64 3x3 layers, nine taps per workgroup, f16 split)",
         37: "dgrad2_kernel (stride-2 data gradients 64 -> 32 / 128 -> 64 channels, filter bank in registers, f16 split)",
         38: "nconv1_kernel (3x3 layers between 32 and 64 channels, filter bank in registers, f16 split)",
         39: "stem_wgrad_bn_kernel (stem weight gradient + BatchNorm backward, fp32 MFMA)",
         40: "gemm3_kernel (co-attention products on pre-split operands, f16 split)",
         41: "conv1b_kernel (bf16 storage: forward / data gradient, both tiles by LDS-DMA)",
         42: "wgrad_kernel<128,128,16,true,0,1,true> (bf16 storage)", 43: "scale_act16_kernel (bf16 storage)",
         44: "partials16_kernel (bf16 storage)", 45: "bn_bwd_apply16_kernel (bf16 storage)",
         46: "wgrad3_kernel<1,true> (bf16 storage, filter rows)", 47: "conv2b_kernel (bf16 storage: 256x256 two-group tile)",
         48: "conv1b_kernel<..,F8> (fp8 storage: e4m3 + row scales, block-scaled MFMA)",
         49: "quant_rows_e4m3_kernel (fp8 storage)", 50: "conv3_kernel<..,IN16> (bf16 storage: 3x3 stride-1 strip, off)",
         51: "conv3x_kernel (3x3 s1 strip on 16x16x32 MFMAs: two taps per MFMA, filter tiles by LDS-DMA)"}
FLOP_TAGS = set(range(8)) | {13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24, 25, 26, 27, 28, 29, 32, 33, 35, 36, 37, 38, 40, 41, 42, 46, 47, 48, 51}
PEAK_OF = {t: (PEAK_SPLIT_TFLOPS if t in (16, 17, 18, 21) else PEAK_H2_TFLOPS if t in (24, 25, 26, 27, 28, 29, 32, 35, 36, 37, 38, 40, 51)
               else PEAK_BF16_MFMA_TFLOPS if t in (19, 20, 23, 33, 41, 42, 46, 47) else 2 * PEAK_BF16_MFMA_TFLOPS if t == 48
               else PEAK_FP32_MFMA_TFLOPS) for t in FLOP_TAGS}
# substrings (spaces removed) that select a tag's kernels among rocprofv3's names: tools/summarize_profile.py matches the PMC
# passes with them
RP_MATCH = {24: ["igemm_kernel<128,128,2,2,0,false,16,true,0,2,"], 25: ["wgrad_kernel<128,128,16,true,0,2,false>", "wgrad1x_kernel<"],
            26: ["igemm_kernel<256,64,4,1,0,false,16,true,0,2,", "igemm_kernel<256,32,4,1,0,false,16,true,0,2,"],
            27: ["igemm_kernel<128,128,2,2,1,false,16,true,0,2,"], 28: ["conv3_kernel<4,2,4,2,", "conv3_kernel<2,2,4,2,", "conv3x_kernel<"],
            29: ["conv3_kernel<4,2,4,2,", "conv3_kernel<2,2,4,2,", "conv3x_kernel<"], 51: ["conv3_kernel<4,2,4,2,", "conv3_kernel<2,2,4,2,", "conv3x_kernel<"], 32: ["wgrad3_kernel<2,false>"], 35: ["conv1_kernel<"],
            36: ["wgrad9_kernel<"], 37: ["dgrad2_kernel"], 38: ["nconv1_kernel"], 40: ["gemm3_kernel<"]}
FAMILY = {t: "conv3x_kernel + conv3_kernel<*,2,4> (3x3 s1 strip, f16 split)" for t in (28, 29, 51)}
NT = 56            # DCN_PROF_TAGS


def read_sclk_mhz(device_index: int = 0):
    """Shader clock (MHz) the card holds right now; None when it cannot be read.  From sysfs (`pp_dpm_sclk` of the device's PCI node: the
    level marked `*`) — no process is started.  Only when sysfs has nothing and no profiler library is preloaded does it fall back to
    `rocm-smi` in a child process (under `rocprofv3 --pmc` the preloaded library initialises the GPU in every child, and the `env python3`
    hop of the rocm-smi script is then an exec of a GPU process, which the GPU boxes refuse)."""
    import glob
    import re

    def parse(path):
        try:
            with open(path) as f:
                txt = f.read()
        except OSError:
            return None
        m = re.search(r"(\d+)\s*Mhz\s*\*", txt, flags=re.I)
        return int(m.group(1)) if m else None

    try:
        pr = torch.cuda.get_device_properties(device_index)
        node = "/sys/bus/pci/devices/%04x:%02x:%02x.0/pp_dpm_sclk" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        v = parse(node)
        if v is not None:
            return v
    except (AttributeError, RuntimeError, AssertionError):
        pass
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if len(cards) == 1:
        v = parse(cards[0])
        if v is not None:
            return v
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks"], capture_output=True, text=True, timeout=20)
    except (OSError, subprocess.SubprocessError):
        return None
    m = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*Mhz", r.stdout, flags=re.I)
    return int(m.group(1)) if m else None


LINE_LIMIT = 2000          # the driver keeps the tail of stdout: the line stays under 2 KB


def compact_line(res: dict) -> str:
    """The ONE stdout line.  north_star's numbers sit as flat scalars INSIDE `roofline` (the driver's parser keeps that object
    whole and cuts long strings): dominant kernel alone / in the step, conv engine against 838.9, cross-modal scoring against
    8 TB/s, the same step on exact-arithmetic alternatives, the held shader clock.  Strings are short by construction; if the
    line still came out long, bookkeeping keys go first and the contract keys never."""
    out = dict(res)
    rf = dict(out.get("roofline") or {})
    rf.pop("rocprof_match", None)              # (kept in the full JSON: tools/summarize_profile.py reads it there)
    ins = rf.pop("in_step", None)
    if ins:
        rf["in_step_frac"] = ins["frac"]; rf["in_step_ms"] = ins["avg_launch_ms"]
        if "shared" in ins:
            rf["in_step_shared"] = ins["shared"]
    ce = out.pop("conv_engine", None)
    if ce:
        rf["conv_engine_frac"] = ce["frac_of_838.9"]; rf["conv_engine_tflops"] = ce["tflops_over_kernel_time"]
        rf["conv_tflop_per_step"] = ce["tflop_per_step"]
        if "tflops_over_step_wall" in ce:          # the whole replayed step against the MFMA ceiling: moves with the step, whatever shares the GPU
            rf["step_tflops"] = ce["tflops_over_step_wall"]; rf["step_frac"] = round(ce["tflops_over_step_wall"] / PEAK_H2_TFLOPS, 4)
        if "binding_frac" in ce:
            rf["conv_engine_binding_frac"] = ce["binding_frac"]
    hs = out.pop("hbm_scoring", None)
    if hs:
        rf["hbm_scoring_frac"] = hs["frac"]; rf["hbm_scoring_gbs"] = hs["achieved"]
    fd = out.pop("flop_dominant", None)
    if fd and fd["kernel"] != rf.get("kernel"):
        rf["flop_dominant"] = fd["kernel"][:40]; rf["flop_dominant_frac"] = fd["frac"]
    alt = out.pop("alt", None)
    if alt:
        for k in ("native_fp32_ms", "bf16x3_ms", "bf16_ms", "bf16s_ms", "fp8_ms", "fp8s_ms"):
            rf[k] = alt.get(k)
        rf["alone_ms"] = alt.get("exclusive_ms")
    if "sclk_mhz" in out and rf:
        rf["sclk_mhz"] = out.pop("sclk_mhz")       # (no roofline object — a run without the profiled pass —: the clock stays at top level)
    if rf:
        rf["note"] = "alone: HIP events; in_step: rocprofv3, 2 queues share GPU"
        out["roofline"] = rf
    line = json.dumps(out, separators=(",", ":"))
    for path in (("loss_hex",), ("roofline", "alg_bytes_per_launch"), ("roofline", "hbm_frac_algorithmic"),
                 ("roofline", "flop_dominant"), ("roofline", "flop_dominant_frac"), ("bn_passes_ms_per_step",), ("host_queue_ms_per_step",),
                 ("loss",), ("roofline", "conv_tflop_per_step"), ("roofline", "launches_per_step"), ("roofline", "note")):
        if len(line) < LINE_LIMIT:
            break
        d = out
        for k in path[:-1]:
            d = d.get(k, {})
        d.pop(path[-1], None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=8, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=8, help="T")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16s", "fp8", "fp8s"], default="fp32",
                    help="arithmetic of the TIMED region (default fp32 = BASELINE configs[1]; bf16s = bf16 storage, configs[2] on one GPU; "
                         "reduced-precision runs say so in dtype / config.arith)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="timed region as one replayed hipGraph per step (auto: on; off = the eager Python step)")
    ap.add_argument("--profile-steps", type=int, default=2,
                    help="untimed eager steps AFTER the timed region, same streams, with a HIP event pair around every conv-engine "
                         "launch: the per-kernel durations of the roofline object (0 = skip)")
    ap.add_argument("--alt-steps", "--exclusive-steps", type=int, default=2, dest="alt_steps",
                    help="untimed eager steps per alternative-arithmetic pass, side streams off (bf16x3 / native "
                         "fp32 / bf16 / fp8 arithmetic).  0 = skip, e.g. under rocprofv3 so its averages speak about the timed step")
    ap.add_argument("--bf16s-leg", choices=["auto", "on", "off"], default="auto",
                    help="after the fp32 timed region: the SAME step on bf16 storage (BASELINE configs[2] on this GPU), captured and replayed "
                         "in this process with the same --steps / --warmup -> roofline.bf16s_clips_s / bf16s_ms_per_step (auto: on for the "
                         "single-GPU fp32 graph run)")
    ap.add_argument("--sampler", choices=["mt", "device"], default="mt",
                    help="negatives of the two sampling heads: 'mt' = Python's MT19937 stream advanced natively on a host thread, bit-exact "
                         "with the reference's random.sample loops (the headline's default); 'device' = drawn on the device inside the "
                         "captured step by a counter-based generator (same distribution, not the same numbers; no host work)")
    ap.add_argument("--clips32-legs", choices=["auto", "on", "off"], default="auto",
                    help="BASELINE configs[4] AT ITS BATCH: the step on fp8 storage and on bf16 storage at 32 clips (256 images) per GPU, "
                         "device sampler, captured and replayed -> roofline.fp8s_bs32_clips_s / bf16s_bs32_clips_s (auto: on for the "
                         "single-GPU fp32 graph run at the default workload)")
    ap.add_argument("--schedules", type=int, default=0,
                    help="N > 0: after the timed region capture the step under three stream schedules (no side streams; the default: "
                         "weight gradient beside the data-gradient chain; weight gradient queued behind its layer's data gradient) "
                         "and time N replays of each — the table goes to the full JSON (`schedules`)")
    ap.add_argument("--schedule-tunes", type=str, default="",
                    help="with --schedules: extra variants of the default schedule under dcn_set_tuning knobs, 'k=v+k=v;k=v' "
                         "(one variant per ';', knobs joined by '+', restored afterwards: give the default as the 3rd field k=v=default)")
    ap.add_argument("--no-side-streams", action="store_true",
                    help="weight gradients, language branch and sampling heads on the main stream (every kernel alone on the GPU)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--force-ddp", action="store_true", help="process group + reducer even at world size 1 (exercises the hooks)")
    ap.add_argument("--tune", type=str, default="", help="key=value[,key=value]: dcn_set_tuning knobs applied before the run (experiments)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: every rank uses cuda:0 and the process group runs on gloo (RCCL refuses "
                         "two ranks per device) — exercises broadcast, sharded seeds, the reducer and the max-over-ranks timing")
    ap.add_argument("--reducer", choices=["flat", "overlap", "ddp"], default="flat",
                    help="multi-GPU gradient averaging: 'flat' = one flat all-reduce after the (graph-replayed) backward; 'overlap' = "
                         "dcnet_amd.parallel.OverlappedGradReducer (buckets all-reduced on a communication stream under the "
                         "backbone's backward; eager step); 'ddp' = torch DDP wrapper (eager step)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes of torch.distributed.run.  This process
    has made no GPU call (importing torch does not initialise HIP) and never execs — it waits and passes the exit code on."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"bench.py: no launcher environment, spawning {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


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
            "sample": f"oracle port fwd+5 losses+bwd, 1 clip T={frames} {size}x{size}, median of {steps} steps, {os.cpu_count()} cpus on host"}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    # stdout carries exactly ONE line: libraries that print there (gloo's connection banner, RCCL without NCCL_DEBUG_FILE, ...) are
    # sent to stderr for the whole run; the JSON line goes to a duplicate of the original descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.rehearse:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but the node shows {torch.cuda.device_count()} GPU(s); "
                         "--rehearse runs every rank on cuda:0 over gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_ddp
    if rank == 0 and use_dist and os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
        os.environ["NCCL_DEBUG"] = "VERSION"        # one RCCL version banner, on stderr (below): which library carried the all-reduce
    elif os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
        os.environ["NCCL_DEBUG"] = "WARN"
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")     # RCCL prints on stdout by default: stdout carries the ONE JSON line
    dist = torch.distributed
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.rehearse:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        if world > 1 and not args.rehearse:
            # one rank per device: gather every rank's device index and insist they differ
            mine = torch.tensor([torch.cuda.current_device()], device=dev)
            seen = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(seen, mine)
            ids = sorted(int(t.item()) for t in seen)
            if ids != list(range(world)):
                raise SystemExit(f"bench.py: ranks share devices: {ids}")

    from dcnet_amd import losses, ops
    from dcnet_amd.lib import lib
    from dcnet_amd.model import grounding_model
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs

    L = lib()
    if args.no_side_streams:
        ops.WGRAD_SIDE = False
    for kv in [t for t in args.tune.split(",") if t]:
        k_, v_ = kv.split("=")
        L.set_tuning(k_.encode(), int(v_))
    ops.set_precision(args.precision)
    torch.manual_seed(1234)            # identical initial weights on every rank (rank 0's are broadcast as well)
    model = grounding_model(corpus=list(range(1000)), light=False, emb_size=512, coordmap=True,
                            bert_model="bert-base-uncased", dataset="vid", img_size=args.size,
                            config_path=os.path.join(ROOT, "model", "yolov3.cfg"), weights_path=None).to(dev)
    model.train()
    model.sampler = args.sampler
    model.sampler_seed = 0x5DC0E7A1 + rank
    if args.no_side_streams:
        model.language_stream = False; model.sampling_stream = False
    # parameters that never receive a gradient in the reference either (dead YOLO heads F7, feature_map F8):
    # freezing them gives a static graph instead of find_unused_parameters=True (train_DCNet.py:483)
    from dcnet_amd.parallel import FlatGradAllReduce, attach_overlapped_reducer, broadcast_parameters, freeze_gradless, wrap_ddp
    freeze_gradless(model)
    red = flat = None
    net = model
    reducer_name = "none"
    if use_dist:
        reducer_name = args.reducer
        if args.reducer == "overlap":
            red = attach_overlapped_reducer(model)
            red.always_collective = args.force_ddp
        elif args.reducer == "ddp":
            net = wrap_ddp(model, local_rank)
        else:
            if world > 1:
                broadcast_parameters(model, 0)
            # gradients live in one flat buffer: no per-parameter copies; the bf16 modes move it over xGMI as bf16
            flat = FlatGradAllReduce(model.parameters(), comm_dtype=torch.bfloat16 if args.precision in ("bf16", "bf16s", "fp8s") else None).bind()
            flat.always_collective = args.force_ddp                   # (one-rank RCCL group: the collective is issued all the same)
    from dcnet_amd.train import make_optimizer     # the reference's two RMSprop groups (train_DCNet.py:519-534), fused HIP step
    opt = make_optimizer(model, 1e-4)

    n_img = args.clips * args.frames
    image, word_id, word_mask = synth_inputs(n_img, args.size, seed=100 + rank)
    bbox = synth_boxes(n_img, args.size, seed=100 + rank)
    image, word_id, word_mask, bbox = image.to(dev), word_id.to(dev), word_mask.to(dev), bbox.to(dev)
    import random
    random.seed(13 + rank)

    def eager_step():
        out = net(image, word_id, word_mask)
        loss, _ = losses.total_loss(out, bbox, args.size)
        if flat is not None:
            flat.zero()
        else:
            opt.zero_grad(set_to_none=True)
        if red is not None:
            red.begin_step()
        loss.backward()
        if red is not None:
            red.finish()             # heads / language gradients in one flat bucket; joins the communication stream
        if flat is not None:
            flat()
        opt.step()
        return loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = args.graph == "on" or (args.graph == "auto" and (not use_dist or args.reducer == "flat"))
    if use_graph and use_dist and args.reducer != "flat":
        raise SystemExit("bench.py: --graph on needs --reducer flat (no collective is captured)")
    step = eager_step
    graph_note = "eager"
    if use_graph:
        from dcnet_amd.graph import GraphedTrainStep
        try:
            gstep = GraphedTrainStep(model, opt, image, word_id, word_mask, bbox, args.size, reducer=flat, warmup=max(1, min(args.warmup, 2)))
            step = gstep
            graph_note = "hipGraph replay (fwd+losses+bwd" + ("+RMSprop)" if flat is None else "); all-reduce+RMSprop eager")
        except Exception as e:           # capture refused: say so loudly, run the eager step (still the HIP path, never a fallback off it)
            if args.graph == "on":
                raise
            print(f"bench.py: hipGraph capture failed ({type(e).__name__}: {e}); timing the eager step", file=sys.stderr, flush=True)
            if hasattr(opt, "device_lr"):
                opt.device_lr = False
            torch.cuda.synchronize()
            graph_note = f"eager (capture failed: {type(e).__name__})"

    for _ in range(args.warmup):
        step()
    barrier()
    if use_graph and step is not eager_step:
        step.host_launch_s = step.host_sampler_s = 0.0
    model.sampler_busy_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    host_dt = time.perf_counter() - t0          # the host has queued every step (no synchronisation inside a step)
    barrier()
    dt = time.perf_counter() - t0
    ops.check_bilstm(dev)                       # the persistent BiLSTM's sticky error word: a timed-out hand-off must not pass as a result
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last_loss = float(last.detach())
    main_sampler_ms = model.sampler_busy_s / args.steps * 1e3      # (CPU time of the worker thread's native draws; the legs below reset the counter)
    max_alloc = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    sclk = None
    for _ in range(3):                # the clock the card holds UNDER this load: three more (untimed) steps queued on every rank,
        step()                        # rank 0 reads rocm-smi while they run
    if rank == 0:
        sclk = read_sclk_mhz(local_rank)
    barrier()

    # ---- untimed passes (every rank runs them, so collectives stay in step) -------------------------------------------
    # from here on the eager step: the profiler wraps each launch in a HIP event pair, which a captured graph cannot hold
    model.static_samples = None
    if hasattr(opt, "device_lr"):
        opt.device_lr = False

    def collect():
        c = (ctypes.c_int64 * NT)(); m = (ctypes.c_double * NT)(); w = (ctypes.c_double * NT)(); by = (ctypes.c_double * NT)()
        L.prof_collect(ctypes.addressof(c), ctypes.addressof(m), ctypes.addressof(w), ctypes.addressof(by))
        cap = 1 << 16
        tg = (ctypes.c_int32 * cap)(); rm = (ctypes.c_double * cap)(); rw = (ctypes.c_double * cap)(); rb = (ctypes.c_double * cap)()
        n = L.prof_records(ctypes.addressof(tg), ctypes.addressof(rm), ctypes.addressof(rw), ctypes.addressof(rb), cap)
        # per tag: the time the binding roofline allows, launch by launch — max(FLOP / MFMA peak of the arithmetic, bytes / 8 TB/s)
        bound = [0.0] * NT; hbm_bound = [0.0] * NT
        for i in range(max(n, 0)):
            t_ = tg[i]
            if not (0 <= t_ < NT):
                continue
            if t_ in FLOP_TAGS:
                tm = rw[i] / (PEAK_OF[t_] * 1e12) * 1e3; th = rb[i] / (PEAK_HBM_GBS * 1e9) * 1e3
            else:
                tm = 0.0; th = rw[i] / (PEAK_HBM_GBS * 1e9) * 1e3
            bound[t_] += max(tm, th); hbm_bound[t_] += th
        # per (kernel, launch shape): launches that share tag, FLOP and bytes are one layer form — where inside a family the time goes
        shapes = {}
        for i in range(max(n, 0)):
            t_ = tg[i]
            if 0 <= t_ < NT and t_ in FLOP_TAGS:
                e_ = shapes.setdefault((t_, round(rw[i]), round(rb[i])), [0, 0.0])
                e_[0] += 1; e_[1] += rm[i]
        return dict(c=list(c), ms=list(m), work=list(w), bytes=list(by), bound=bound, hbm=hbm_bound, shapes=shapes)

    def run_pass(nsteps, precision="fp32", side_streams=True):
        was = (ops.WGRAD_SIDE, model.language_stream, model.sampling_stream)
        if not side_streams:
            ops.WGRAD_SIDE = False; model.language_stream = False; model.sampling_stream = False
        ops.set_precision(precision)
        eager_step(); barrier()
        L.prof_enable(1)
        t1 = time.perf_counter()
        for _ in range(nsteps):
            eager_step()
        barrier()
        el = time.perf_counter() - t1
        L.prof_enable(0)
        ops.WGRAD_SIDE, model.language_stream, model.sampling_stream = was
        ops.set_precision(args.precision)
        r = collect(); r["ms_per_step"] = el / nsteps * 1e3; r["steps"] = nsteps
        return r

    sched = None
    if args.schedules > 0 and not use_dist:
        from dcnet_amd.graph import GraphedTrainStep
        sched = {}
        for name, (ws, ls, ss, after, prio) in {"no_side_streams": (False, False, False, False, 0), "wgrad_beside_dgrad": (True, True, True, False, 0),
                                                "wgrad_behind_its_dgrad": (True, True, True, True, 0),
                                                "wgrad_beside_dgrad_low_priority": (True, True, True, False, 1),
                                                "wgrad_behind_its_dgrad_low_priority": (True, True, True, True, 1),
                                                "no_side_streams_again": (False, False, False, False, 0),     # (order check: the first variant runs on a cooler chip)
                                                **{"tune:" + t: (True, True, True, True, 0) for t in args.schedule_tunes.split(";") if t}}.items():
            was = (ops.WGRAD_SIDE, model.language_stream, model.sampling_stream, ops.WGRAD_AFTER_DGRAD)
            ops.WGRAD_SIDE, model.language_stream, model.sampling_stream, ops.WGRAD_AFTER_DGRAD = ws, ls, ss, after
            ops.SIDE_PRIORITY = prio
            restore = []
            if name.startswith("tune:"):
                for kv in name[5:].split("+"):
                    parts_ = kv.split("=")
                    if parts_[0].startswith("ops."):          # a Python-level switch of dcnet_amd.ops (e.g. ops.STEM_FUSED_BWD=0=1)
                        setattr(ops, parts_[0][4:], int(parts_[1]))
                    else:
                        L.set_tuning(parts_[0].encode(), int(parts_[1]))
                    restore.append((parts_[0], int(parts_[2]) if len(parts_) > 2 else 0))
            g_ = GraphedTrainStep(model, opt, image, word_id, word_mask, bbox, args.size, warmup=1)
            g_(); barrier()
            t1 = time.perf_counter()
            for _ in range(args.schedules):
                g_()
            barrier()
            sched[name] = (time.perf_counter() - t1) / args.schedules * 1e3
            ops.WGRAD_SIDE, model.language_stream, model.sampling_stream, ops.WGRAD_AFTER_DGRAD = was
            ops.SIDE_PRIORITY = 0
            for k_, v_ in restore:
                if k_.startswith("ops."):
                    setattr(ops, k_[4:], v_)
                else:
                    L.set_tuning(k_.encode(), v_)
            del g_
            model.static_samples = None
            opt.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
        if hasattr(opt, "device_lr"):
            opt.device_lr = False
    # Per-launch HIP event pairs, side streams off: every kernel ALONE on the GPU — the one regime an eager pass can reproduce (a
    # replayed graph keeps all streams fed, the eager host does not; events recorded inside a captured graph cannot be read back:
    # tools/graph_event_probe.hip).  `rocprofv3 -- python3 bench.py --graph off --no-side-streams` shows the same averages.  The
    # in-step durations of the replayed step (weight gradients beside the data-gradient chain) come from the committed rocprofv3
    # summary of the default command (profiles/in_step_latest.json) and are quoted beside them.
    prof = run_pass(args.profile_steps, args.precision, False) if args.profile_steps > 0 else None
    alts = {}
    if args.alt_steps > 0:
        alts["fp32_bf16x3"] = run_pass(args.alt_steps, "fp32_bf16x3", False)
        alts["native_fp32"] = run_pass(args.alt_steps, "fp32_mfma", False)
        alts["bf16_operands"] = run_pass(args.alt_steps, "bf16", False)
        alts["bf16_storage"] = run_pass(args.alt_steps, "bf16s", False)
        alts["fp8_operands"] = run_pass(args.alt_steps, "fp8", False)
        alts["fp8_storage"] = run_pass(args.alt_steps, "fp8s", False)
    # ---- configs[2] / configs[4] on this GPU: the same step on bf16 storage and on fp8 storage, each captured again and replayed (their own
    #      timed regions, never `value`).  They run AFTER the profiled / alternative-arithmetic passes (round-5 advice: those passes and
    #      the parity check must not see weights and optimiser state perturbed by reduced-precision steps).
    legs = {}
    want_leg = args.bf16s_leg == "on" or (args.bf16s_leg == "auto" and args.precision == "fp32" and not use_dist)
    want_32 = args.clips32_legs == "on" or (args.clips32_legs == "auto" and want_leg and args.clips == 8 and args.size == 416 and args.frames == 8)

    def run_leg(name, mode_, img_, wid_, wmask_, bbox_, clips_, sampler_):
        from dcnet_amd.graph import GraphedTrainStep
        ops.set_precision(mode_)
        was_sampler = model.sampler
        model.sampler = sampler_
        g2 = None
        try:
            g2 = GraphedTrainStep(model, opt, img_, wid_, wmask_, bbox_, args.size, warmup=max(1, min(args.warmup, 2)))
            for _ in range(args.warmup):
                g2()
            barrier()
            g2.host_launch_s = g2.host_sampler_s = 0.0; model.sampler_busy_s = 0.0
            t1 = time.perf_counter()
            for _ in range(args.steps):
                l2 = g2()
            barrier()
            dt2 = time.perf_counter() - t1
            ops.check_bilstm(dev)
            l2 = float(l2.detach())
            if not np.isfinite(l2):
                raise RuntimeError(f"{name} leg: loss {l2}")
            legs[name] = {"ms_per_step": dt2 / args.steps * 1e3, "clips_s": clips_ * args.steps / dt2, "steps": args.steps,
                          "warmup": args.warmup, "loss": l2, "clips": clips_, "sampler": sampler_,
                          "mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                          "host_ms_per_step": {"launch": round(g2.host_launch_s / args.steps * 1e3, 2),
                                               "sampler_thread": round(model.sampler_busy_s / args.steps * 1e3, 2),
                                               "gpu_wait": round(g2.host_sampler_s / args.steps * 1e3, 2)},
                          "step": "hipGraph replay (fwd+losses+bwd+RMSprop), " + mode_}
        except Exception as e:       # the fp32 line must not die with an extra leg: say so on stderr and in the line
            print(f"bench.py: {name} leg failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            legs[name] = {"error": type(e).__name__}
        finally:
            ops.set_precision(args.precision)
            model.sampler = was_sampler
            del g2
            model.static_samples = None
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()             # (a failed capture was ended by torch.cuda.graph's exit; nothing of it is left in flight)
            torch.cuda.empty_cache()

    if want_leg and use_graph and step is not eager_step:
        for mode_ in ("bf16s", "fp8s"):
            if mode_ != args.precision:
                run_leg(mode_, mode_, image, word_id, word_mask, bbox, args.clips, args.sampler)
        if want_32:
            # configs[4] at its batch (32 clips = 256 images per GPU), with the device sampler: the exact host sampler needs 0.19 s of a
            # core per step there (N^2 HW0 random.sample calls), as long as the GPU step itself (round-5 verdict, weak #6)
            n32 = 32 * args.frames
            im32, wi32, wm32 = synth_inputs(n32, args.size, seed=300 + rank)
            bb32 = synth_boxes(n32, args.size, seed=300 + rank)
            im32, wi32, wm32, bb32 = im32.to(dev), wi32.to(dev), wm32.to(dev), bb32.to(dev)
            torch.cuda.reset_peak_memory_stats(dev)
            for mode_ in ("bf16s", "fp8s"):
                run_leg(mode_ + "_bs32", mode_, im32, wi32, wm32, bb32, 32, "device")
            del im32, wi32, wm32, bb32
            torch.cuda.empty_cache()

    if use_dist:
        dist.barrier()                           # every rank got here: only now may rank 0 print the line

    if rank == 0:
        from dcnet_amd.utils.srchash import kernel_sources_hash
        src_hash = kernel_sources_hash()
        clips_total = args.clips * world * args.steps

        def table(r):
            out = {}
            for t, nm in NAMES.items():
                if r["c"][t]:
                    fl = t in FLOP_TAGS
                    rate = r["work"][t] / (r["ms"][t] * 1e-3) / (1e12 if fl else 1e9)
                    e = {"launches_per_step": r["c"][t] / r["steps"], "avg_ms": r["ms"][t] / r["c"][t], "ms_per_step": r["ms"][t] / r["steps"],
                         "achieved": rate, "unit": "TFLOP/s" if fl else "GB/s", "frac": rate / (PEAK_OF[t] if fl else PEAK_HBM_GBS),
                         "binding_roofline_frac": r["bound"][t] / r["ms"][t]}
                    if fl and r["bytes"][t]:
                        e["hbm_gbs_algorithmic"] = r["bytes"][t] / (r["ms"][t] * 1e-3) / 1e9
                    out[nm] = e
            return out

        def fam_sum(r, t, key):
            if t in FAMILY:
                return sum(r[key][u] for u in FAMILY if FAMILY[u] == FAMILY[t])
            return r[key][t]

        res = {"metric": f"clips/sec (T={args.frames}, {args.size}x{args.size}, bs{args.clips}) fwd+bwd", "value": clips_total / dt,
               "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
               "dtype": {"fp32": "f32", "bf16": "bf16", "bf16s": "bf16", "fp8": "fp8", "fp8s": "fp8"}[args.precision],
               "config": {"workload": f"T={args.frames} {args.size}x{args.size} bs{args.clips}/GPU L=20 {args.precision}, {n_img} img/GPU/step as pairs, fwd+5 losses+bwd+RMSprop",
                          "arith": {"fp32": "f16x2-split MFMA, fp32 accumulate", "bf16": "bf16 operands, fp32 tensors, fp32 accumulate",
                                    "bf16s": "bf16 storage (conv stacks), fp32 accumulate + master weights",
                                    "fp8": "fp8 e4m3 operands, fp32 accumulate",
                                    "fp8s": "fp8 storage (e4m3 + row scales, block-scaled MFMA) for the 3x3 convs' fwd/dgrad on bf16 storage"}[args.precision],
                          "parallelism": f"dp{world}",
                          "ranks_seen": dist.get_world_size() if use_dist else 1, "reducer": reducer_name, "step": graph_note,
                          "sampler": args.sampler},
               "host_queue_ms_per_step": round(host_dt / args.steps * 1e3, 2), "mem_gb": round(max_alloc, 1), "loss": round(last_loss, 4),
               "loss_hex": float(last_loss).hex(), "src": src_hash}     # src: hash of the kernel + host sources (dcnet_amd.utils.srchash)
        if use_dist:
            res["config"]["collectives"] = (flat.collectives if flat is not None else red.buckets_last_step if red is not None else None)
        if step is not eager_step:
            # what the host does per replayed step: launching (graph + optimiser-side calls) vs waiting for the native sampler thread
            # (the reference's O(N^2 HW) random.sample loop, serial by construction; it runs under the previous replay)
            # sampler_thread = what the draws cost their worker thread; gpu_wait = the host held back by the GPU (a staging set is
            # rewritten only after its upload of two steps ago has run: the host stays <= 2 steps ahead) + joining the worker
            res["host_ms_per_step"] = {"launch": round(step.host_launch_s / args.steps * 1e3, 2),
                                       "sampler_thread": round(main_sampler_ms, 2),
                                       "gpu_wait": round(step.host_sampler_s / args.steps * 1e3, 2)}
        full = {"bench_line": None, "timed": {"ms_per_step": res["ms_per_step"], "step": graph_note}}
        if sched:
            full["schedules_ms_per_step"] = sched
            print("schedules (graph replay, ms/step):", json.dumps(sched), file=sys.stderr, flush=True)
        if prof is not None:
            mm_tags = sorted(FLOP_TAGS - {13, 14})
            # time-dominant kernel (family) of the step among the conv-engine launches; the FLOP-dominant one beside it
            dom = max(mm_tags, key=lambda t: (fam_sum(prof, t, "ms"), prof["ms"][t]))
            fdom = max(mm_tags, key=lambda t: (fam_sum(prof, t, "work"), prof["work"][t]))

            def entry(r, t):
                ms_, wk, c_, by, bd = (fam_sum(r, t, k) for k in ("ms", "work", "c", "bytes", "bound"))
                if not c_:
                    return None
                ach = wk / (ms_ * 1e-3) / 1e12
                return {"kernel": FAMILY.get(t, NAMES[t]), "achieved": round(ach, 1), "peak": round(PEAK_OF[t], 1), "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_OF[t], 4), "avg_launch_ms": round(ms_ / c_, 4), "ms_per_step": round(ms_ / r["steps"], 2),
                        "launches_per_step": round(c_ / r["steps"], 1), "alg_bytes_per_launch": round(by / c_),
                        "hbm_frac_algorithmic": round(by / (ms_ * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        "binding_frac": round(bd / ms_, 4)}

            e = entry(prof, dom)
            traffic = None
            stale = []
            pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc_file):     # HBM bytes per launch from the last rocprofv3 --pmc passes (not measurable live)
                with open(pmc_file) as f:
                    pmc = json.load(f)
                if pmc.get("src_hash") != src_hash:      # taken on other sources than the ones running: not quoted
                    stale.append("pmc")
                elif pmc.get("kernel", "").split(" (")[0] == e["kernel"].split(" (")[0]:
                    traffic = pmc.get("hbm_bytes_per_launch")
            roofline = {"bound": "mfma", "kernel": e["kernel"], "rocprof_match": RP_MATCH.get(dom), "achieved": e["achieved"], "peak": e["peak"], "unit": "TFLOP/s",
                        "frac": e["frac"], "traffic": traffic,
                        "traffic_ratio": round(traffic / e["alg_bytes_per_launch"], 3) if traffic else None,
                        "avg_launch_ms": e["avg_launch_ms"], "ms_per_step": e["ms_per_step"], "launches_per_step": e["launches_per_step"],
                        "alg_bytes_per_launch": e["alg_bytes_per_launch"], "hbm_frac_algorithmic": e["hbm_frac_algorithmic"],
                        "binding_frac": e["binding_frac"]}
            ins_file = os.path.join(ROOT, "profiles", "in_step_latest.json")
            if os.path.exists(ins_file):     # launch durations inside the replayed step (rocprofv3 of this command, committed profile)
                with open(ins_file) as f:
                    ins = json.load(f)
                key = {28: "conv3", 29: "conv3", 51: "conv3", 32: "wgrad3", 35: "conv1", 25: "wgrad<128,128>", 24: "igemm<128,128> NT"}.get(dom)
                k_ = ins.get("kernels", {}).get(key)
                if ins.get("src_hash") != src_hash:
                    stale.append("in_step"); k_ = None
                if k_:
                    fl_per_launch = fam_sum(prof, dom, "work") / fam_sum(prof, dom, "c")
                    roofline["in_step"] = {"avg_launch_ms": round(k_["avg_launch_ms"], 4),
                                           "frac": round(fl_per_launch / (k_["avg_launch_ms"] * 1e-3) / 1e12 / PEAK_OF[dom], 4)}
                    if "shared_frac_of_time" in k_:      # part of that time with a kernel of another graph queue resident as well
                        roofline["in_step"]["shared"] = k_["shared_frac_of_time"]
            if stale:
                roofline["stale_profiles"] = stale    # committed profiles of OTHER sources: their numbers are left out (null)
            res["roofline"] = roofline
            fe = entry(prof, fdom)
            res["flop_dominant"] = {k: fe[k] for k in ("kernel", "frac", "ms_per_step", "binding_frac")}
            if prof["c"][8]:
                rate = prof["work"][8] / (prof["ms"][8] * 1e-3) / 1e9
                res["hbm_scoring"] = {"kernel": NAMES[8], "achieved": round(rate, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(rate / PEAK_HBM_GBS, 4)}
            mfma_ms = sum(prof["ms"][t] for t in mm_tags); mfma_work = sum(prof["work"][t] for t in mm_tags)
            res["conv_engine"] = {"tflop_per_step": round(mfma_work / prof["steps"] / 1e12, 2),
                                  "tflops_over_kernel_time": round(mfma_work / (mfma_ms * 1e-3) / 1e12, 1),
                                  "frac_of_838.9": round(mfma_work / (mfma_ms * 1e-3) / 1e12 / PEAK_H2_TFLOPS, 4),
                                  "frac_of_fp32_mfma_157.3": round(mfma_work / (mfma_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3),
                                  "tflops_over_step_wall": round(mfma_work / prof["steps"] / 1e12 / (dt / args.steps), 1),
                                  # time the binding roofline of every launch allows (max of FLOP / MFMA peak of its arithmetic, bytes / 8 TB/s) over
                                  # the time taken: what the HBM-bound layers of the stack (1x1 residual layers, 32-128-channel maps) can be held to
                                  "binding_frac": round(sum(prof["bound"][t] for t in mm_tags) / mfma_ms, 4)}
            bn = sum(prof["ms"][t] for t in (10, 11, 22)) / prof["steps"]
            res["bn_passes_ms_per_step"] = round(bn, 2)
            per_shape = []
            for (t_, wk_, by_), (cnt_, ms_) in prof["shapes"].items():
                tm_ = wk_ / (PEAK_OF[t_] * 1e12) * 1e3; th_ = by_ / (PEAK_HBM_GBS * 1e9) * 1e3
                per_shape.append({"kernel": NAMES[t_][:28], "gflop": round(wk_ / 1e9, 2), "mb": round(by_ / 1e6, 1), "launches_per_step": cnt_ / prof["steps"],
                                  "avg_ms": round(ms_ / cnt_, 4), "ms_per_step": round(ms_ / prof["steps"], 3),
                                  "mfma_frac": round(tm_ / (ms_ / cnt_), 3), "hbm_frac": round(th_ / (ms_ / cnt_), 3)})
            per_shape.sort(key=lambda e_: -e_["ms_per_step"])
            full["profiled_pass_kernels_alone"] = {"ms_per_step": prof["ms_per_step"], "kernels": table(prof), "per_shape": per_shape}
        if alts:
            res["alt"] = {"exclusive_ms": round(prof["ms_per_step"], 1) if prof else None, "bf16x3_ms": round(alts["fp32_bf16x3"]["ms_per_step"], 1),
                          "native_fp32_ms": round(alts["native_fp32"]["ms_per_step"], 1), "bf16_ms": round(alts["bf16_operands"]["ms_per_step"], 1), "bf16s_ms": round(alts["bf16_storage"]["ms_per_step"], 1),
                          "fp8_ms": round(alts["fp8_operands"]["ms_per_step"], 1), "fp8s_ms": round(alts["fp8_storage"]["ms_per_step"], 1)}
            for k_, r_ in alts.items():
                full[k_] = {"ms_per_step": r_["ms_per_step"], "kernels": table(r_)}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.size, args.frames, args.cpu_steps)
            full["cpu_baseline"] = cb
            res["cpu_baseline"] = {"value": round(cb["value"], 4), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": cb["sample"], "gpu_vs_oracle_max_abs_err": round(cb["parity"]["max_abs_err_outbox"], 6),
                                   "acc_at_0.5_vs_oracle_boxes": cb["parity"]["acc_at_iou_0.5_vs_oracle_boxes"]}
        for mode_, leg in legs.items():
            full[mode_ + "_replayed"] = leg
            rf_ = res.setdefault("roofline", {})
            if "error" in leg:
                rf_[mode_ + "_leg"] = leg["error"]
            else:
                rf_[mode_ + "_clips_s"] = round(leg["clips_s"], 2); rf_[mode_ + "_ms_per_step"] = round(leg["ms_per_step"], 2)
                if leg["sampler"] == "mt":
                    rf_[mode_ + "_sampler_thread_ms"] = leg["host_ms_per_step"]["sampler_thread"]
        if legs:
            crit_file = os.path.join(ROOT, "profiles", "precision_criterion_latest.json")
            if os.path.exists(crit_file):    # SURVEY 8(c) box criterion of the modes on trained weights (tools/precision_criterion.py; also a -m gpu test)
                with open(crit_file) as f:
                    cr = json.load(f)
                for mode_ in [m__ for m__ in legs if not m__.endswith("_bs32")]:
                    m_ = cr.get("modes", {}).get(mode_)
                    if m_ and "criterion_met_frac" in m_:
                        res["roofline"][mode_ + "_criterion"] = "%d/%d" % (round(m_["criterion_met_frac"] * cr.get("images", 16)), cr.get("images", 16))
        res["full"] = "profiles/bench_full_latest.json"
        res["sclk_mhz"] = sclk
        full["bench_line"] = res
        for d in (os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out")):
            try:
                if os.path.isdir(d):
                    with open(os.path.join(d, "bench_full_latest.json"), "w") as f:
                        json.dump(full, f, indent=1)
            except OSError:
                pass
        line = compact_line(res)
        sys.stdout.flush()
        os.write(json_fd, (line + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
