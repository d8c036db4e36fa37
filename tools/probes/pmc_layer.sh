# usage: bash tools/pmc_layer.sh "<cin,cout,k,stride,h>" "<counters pass 1>" ["<counters pass 2>" ...]
# per-kernel PMC averages for one conv shape (fwd / dgrad / wgrad), counters in separate passes
set -eo pipefail
ROOT=$(pwd); L=$1; shift
OUT=$ROOT/gpurun_out/pmc_layer; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/tools/bench_convs.py --only $L --iters 3 > $OUT/p$i.log 2>&1)
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
    if "igemm" in k or "wgrad" in k or "conv3" in k or "conv1" in k or "nconv" in k or "dgrad2" in k:
        print(k[:110])
        for c, (n, s) in sorted(cs.items()):
            print(f"    {c:34s} n={n:3d} avg={s/n:16.1f}")
PY
