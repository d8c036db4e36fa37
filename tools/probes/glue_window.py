"""Where the step runs no library kernel: python tools/glue_window.py <kernel_trace.csv> [step_ms]
Prints, for the last step of a rocprofv3 kernel trace, the time during which no libdcnet_hip kernel is running on any
queue (torch glue + launch gaps), and the 2.5-ms windows of the main queue with many torch launches."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
step_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 165.0
tend = max(r[1] for r in rows); t0 = tend - step_ms * 1e6
step = [r for r in rows if r[0] >= t0]
is_torch = lambda n: "at::native" in n or n.startswith("Cijk") or "rocclr" in n or "rocprim" in n
byq = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for s, e, n, q in step:
    k = byq[q]
    if is_torch(n): k[0] += 1; k[1] += (e - s) / 1e6
    else: k[2] += 1; k[3] += (e - s) / 1e6
for q, v in sorted(byq.items()):
    print("queue %s: torch %4d launches %6.2f ms | library %4d launches %7.2f ms" % ((q,) + tuple(v)))
lib = sorted((s, e) for s, e, n, q in step if not is_torch(n))
merged = []
for s, e in lib:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
busy = sum(e - s for s, e in merged) / 1e6
span = (max(e for s, e, _, _ in step) - min(s for s, e, _, _ in step)) / 1e6
print("span %.1f ms: a library kernel is running %.1f ms, none %.1f ms" % (span, busy, span - busy))
mainq = max(byq, key=lambda q: byq[q][3])
bins = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for s, e, n, q in step:
    if q != mainq: continue
    k = bins[int((s - t0) / 2.5e6)]
    if is_torch(n): k[0] += 1; k[1] += (e - s) / 1e6
    else: k[2] += 1; k[3] += (e - s) / 1e6
print("main queue windows with >= 20 torch launches:  t(ms)  torch n / ms | library n / ms")
for b in sorted(bins):
    v = bins[b]
    if v[0] >= 20: print("   %6.1f  %4d %5.2f | %4d %5.2f" % (b * 2.5, v[0], v[1], v[2], v[3]))
