# usage: bash tools/pmc_cache.sh "<cin,cout,k,stride,h>"
# cache-hierarchy counters of one conv shape (one counter set per pass; a pass with an unknown counter is skipped)
set -o pipefail
ROOT=$(pwd); L=$1
OUT=$ROOT/gpurun_out/pmc_cache; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TD_BUSY_avr" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/tools/bench_convs.py --only $L --iters 3 > $OUT/p$i.log 2>&1) || echo "pass $i ($C) failed"
done
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1<<30)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); nm = {}
    for r in csv.DictReader(open(f, newline="")):
        k = (r["Dispatch_Id"], r["Counter_Name"]); per[k] += float(r["Counter_Value"]); nm[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per.items():
        a = acc[nm[d]][c]; a[0] += 1; a[1] += v
for k, cs in acc.items():
    if "igemm" in k or "wgrad" in k or "conv3" in k:
        print(k[:110])
        for c, (n, s) in sorted(cs.items()):
            print(f"    {c:34s} n={n:3d} avg={s/n:16.1f}")
PY
