set -eo pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 --profile-steps 0 --bf16s-leg off"
rm -rf "$OUT/stats" "$OUT/stats_bf16s"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_bf16s" -o bench -- python3 $BENCH --precision bf16s > "$OUT/bench_bf16s_under_rocprof.json" 2> "$OUT/stats_bf16s.err"
echo "bf16s stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_fp8s" -o bench -- python3 $BENCH --precision fp8s > "$OUT/bench_fp8s_under_rocprof.json" 2> "$OUT/stats_fp8s.err"
echo "fp8s stats done"
cd "$ROOT"
find "$OUT" -name '*_kernel_trace.csv' -size +20M -delete || true
du -sh "$OUT"
