#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:
#   bash tools/gpu_profile.sh <tag>
# 1. bench.py as the driver runs it            -> gpurun_out/<tag>/bench.json
# 2. the same command under rocprofv3 --stats  -> gpurun_out/<tag>/stats/   (kernel trace only)
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; no other trace domains) -> gpurun_out/<tag>/pmc_*/
# tools/summarize_profile.py turns 2 and 3 into the files committed under profiles/.
set -eo pipefail
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
cp "$ROOT/gpurun_out/bench_full_latest.json" "$OUT/bench_full.json" 2>/dev/null || true
echo "bench done"; tail -c 600 "$OUT/bench.json"; echo
# the default command (graph replays; no eager profiled pass, so every traced step runs with the replay's stream concurrency)
BENCH="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 --profile-steps 0 --bf16s-leg off"      # (no extra legs: the trace's last three steps are this mode's replays)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats done"
# the eager step with the per-launch event pairs (what bench.py's live roofline numbers are taken from): its averages must agree with them
BENCHE="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 --profile-steps 2 --graph off --no-side-streams"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager" -o bench -- python3 $BENCHE > "$OUT/bench_eager_under_rocprof.json" 2> "$OUT/stats_eager.err"
echo "eager stats done"
# the bf16-storage mode (BASELINE configs[2] on one GPU): the timed line, then its graph replays under the tracer
python3 "$ROOT/bench.py" --precision bf16s --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > "$OUT/bench_bf16s.json" 2> "$OUT/bench_bf16s.err"
cp "$ROOT/gpurun_out/bench_full_latest.json" "$OUT/bench_full_bf16s.json" 2>/dev/null || true
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_bf16s" -o bench -- python3 $BENCH --precision bf16s > "$OUT/bench_bf16s_under_rocprof.json" 2> "$OUT/stats_bf16s.err"
echo "bf16s stats done"
# the fp8-storage mode (BASELINE configs[4]): the timed line at batch 8 and at the config's batch 32 (device sampler: the exact host sampler
# needs 0.19 s of a core per step there), and bf16 storage at batch 32 beside it
python3 "$ROOT/bench.py" --precision fp8s --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 > "$OUT/bench_fp8s.json" 2> "$OUT/bench_fp8s.err"
cp "$ROOT/gpurun_out/bench_full_latest.json" "$OUT/bench_full_fp8s.json" 2>/dev/null || true
python3 "$ROOT/bench.py" --precision fp8s --clips 32 --sampler device --steps 5 --warmup 2 --no-cpu-baseline --alt-steps 0 --profile-steps 0 > "$OUT/bench_fp8s_clips32.json" 2> "$OUT/bench_fp8s_clips32.err"
python3 "$ROOT/bench.py" --precision bf16s --clips 32 --sampler device --steps 5 --warmup 2 --no-cpu-baseline --alt-steps 0 --profile-steps 0 > "$OUT/bench_bf16s_clips32.json" 2> "$OUT/bench_bf16s_clips32.err"
echo "fp8s done"
# SURVEY 8(c) box criterion of the reduced-precision modes on trained weights
python3 "$ROOT/tools/precision_criterion.py" --out "$OUT/precision_criterion.json" > "$OUT/precision_criterion.log" 2>&1 || true
echo "criterion done"
BENCH1="$ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --alt-steps 0 --profile-steps 1 --graph off"      # (counter collection serialises dispatches: the eager step)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 $BENCH1 > /dev/null 2> "$OUT/pmc_fetch.err"
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 $BENCH1 > /dev/null 2> "$OUT/pmc_write.err"
echo "pmc write done"
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" --no-copy || true
# keep the merged-back payload small: counter_collection csv can be large, the per-kernel summaries are enough
find "$OUT" -name '*_kernel_trace.csv' -size +20M -delete || true
find "$OUT" -name '*counter_collection.csv' -size +20M -delete || true
du -sh "$OUT"
