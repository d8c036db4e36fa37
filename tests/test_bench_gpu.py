"""bench.py as the driver runs it: ONE JSON line on stdout, short enough to survive a tail capture (< 2 KB), carrying the
roofline / cpu_baseline objects; and `--gpus N` without a launcher spawns its own ranks (rehearsed here with two ranks on the
one GPU of the box over gloo: broadcast, per-rank seeds, flat gradient all-reduce, max-over-ranks timing, barrier before the
line)."""
import json
import os
import subprocess
import sys

import pytest

from util import ROOT

pytestmark = pytest.mark.gpu


def _run(args, timeout=420):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 2048, len(lines[0])
    return json.loads(lines[0]), r.stderr


def test_single_gpu_line_is_compact_and_complete():
    res, _ = _run(["--steps", "2", "--warmup", "1", "--clips", "2", "--frames", "2", "--size", "256", "--alt-steps", "1", "--cpu-steps", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["dtype"] == "f32" and res["vs_baseline"] is None
    assert "workload" in res["config"] and "hipGraph" in res["config"]["step"]
    rf = res["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "ms_per_step", "binding_frac",
              "conv_engine_frac", "hbm_scoring_frac", "native_fp32_ms", "bf16x3_ms", "bf16_ms", "fp8_ms", "alone_ms", "sclk_mhz"):
        assert k in rf, k
    assert rf["sclk_mhz"] is None or 50 <= rf["sclk_mhz"] <= 3000
    assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = res["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["gpu_vs_oracle_max_abs_err"] < 1e-3
    assert abs(res["value"] - 2 * 1000.0 / res["ms_per_step"]) < 1e-6 * res["value"] + 1e-9      # clips / s from the same clock
    assert os.path.exists(os.path.join(ROOT, "profiles", "bench_full_latest.json"))


def test_gpus_2_self_launch_rehearsal():
    """No launcher environment: bench.py must start its own two ranks (children of torch.distributed.run), not exit."""
    res, err = _run(["--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1", "--clips", "1", "--frames", "2", "--size", "256",
                     "--no-cpu-baseline", "--alt-steps", "0", "--profile-steps", "0"])
    assert res["n_gpus"] == 2 and res["config"]["ranks_seen"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["config"]["reducer"] == "flat" and res["scaling"] == "weak"
    assert abs(res["value"] - 2 * 1 * 1000.0 / res["ms_per_step"]) < 1e-6 * res["value"] + 1e-9   # whole-job clips / s
    assert "spawning 2 ranks" in err


def test_one_rank_rccl_group_behind_the_graph_replay():
    """RCCL insurance on one GPU: a one-rank **nccl** process group with the collective forced — the flat all-reduce issued right
    behind a graph replay (the data-parallel default), then the overlapped reducer's buckets on its communication stream.  A sum
    over one rank is the identity: the overlapped reducer (gradients handed to autograd as usual) must leave the loss BITWISE where
    the run without a process group puts it; the flat reducer (gradients accumulated into views of one buffer) within 1e-4 — this
    2-image training is chaotic (an ulp grows a thousandfold per step), so an early step is compared."""
    common = ["--steps", "1", "--warmup", "0", "--clips", "1", "--frames", "2", "--size", "256", "--no-cpu-baseline", "--alt-steps", "0",
              "--profile-steps", "0"]
    base, _ = _run(common)
    flat, err = _run(common + ["--force-ddp", "--reducer", "flat", "--graph", "on"])
    assert flat["config"]["ranks_seen"] == 1 and flat["config"]["reducer"] == "flat" and "hipGraph" in flat["config"]["step"]
    assert flat["config"]["collectives"] >= 1 + 1 + 1 + 3, flat["config"]    # warm-up + capture + timed + the three clock-read steps
    assert abs(flat["loss"] - base["loss"]) < 1e-4 * abs(base["loss"]), (flat["loss_hex"], base["loss_hex"])
    eager = ["--steps", "2", "--warmup", "1"] + common[4:]
    base_e, _ = _run(eager + ["--graph", "off"])                           # (the overlapped reducer runs the eager step)
    over, _ = _run(eager + ["--force-ddp", "--reducer", "overlap"])
    assert over["config"]["ranks_seen"] == 1 and over["config"]["reducer"] == "overlap" and over["config"]["collectives"] >= 2
    assert over["loss_hex"] == base_e["loss_hex"], (over["loss"], base_e["loss"])


def test_bf16_storage_run_with_a_bf16_gradient_buffer_on_the_wire():
    """configs[2] on the one GPU of the box: `--precision bf16s` (bf16 storage) as a timed hipGraph run, and the same behind a one-rank
    RCCL group whose flat all-reduce moves a bf16 copy of the gradient buffer (cast, collective, cast back): the line says what ran,
    the loss stays of the order the un-reduced bf16 run puts it (the gradients are rounded to bf16 on the wire)."""
    common = ["--steps", "1", "--warmup", "0", "--clips", "1", "--frames", "2", "--size", "256", "--no-cpu-baseline", "--alt-steps", "0",
              "--profile-steps", "0", "--precision", "bf16s"]
    base, _ = _run(common)
    assert base["dtype"] == "bf16" and "bf16 storage" in base["config"]["arith"] and "hipGraph" in base["config"]["step"]
    fp32, _ = _run(common[:-2])
    # (this 2-image training is chaotic — an ulp grows a thousandfold per step — so the third step's loss is held to the same order only)
    assert fp32["dtype"] == "f32" and 0.2 < base["loss"] / fp32["loss"] < 5.0 and base["loss_hex"] != fp32["loss_hex"]
    flat, _ = _run(common + ["--force-ddp", "--reducer", "flat", "--graph", "on"])
    assert flat["config"]["reducer"] == "flat" and flat["config"]["collectives"] >= 6 and flat["dtype"] == "bf16"
    assert 0.2 < flat["loss"] / base["loss"] < 5.0, (flat["loss"], base["loss"])
