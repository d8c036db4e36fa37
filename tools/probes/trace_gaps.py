"""Summarise GPU idle gaps from a rocprofv3 kernel_trace CSV: python tools/trace_gaps.py <csv> [min_gap_us] [window_ms]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
if len(sys.argv) > 3:      # analyse only the last <window_ms> of the trace (e.g. the final timed step)
    tend = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= tend - float(sys.argv[3]) * 1e6]
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; cur_end = rows[0][0]; gaps = []
for s, e, n in rows:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e); prev = n
print("span %.1f ms busy %.1f ms idle %.1f ms  kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(rows)))
big = [g for g in gaps if g[0] > thr]
print("gaps > %.0f us: %d totalling %.1f ms; small gaps total %.1f ms" % (thr / 1e3, len(big), sum(g[0] for g in big) / 1e6,
                                                                      sum(g[0] for g in gaps if g[0] <= thr) / 1e6))
for g in sorted(big, reverse=True)[:25]:
    print("  %8.1f us  after %-60s before %s" % (g[0] / 1e3, g[1], g[2]))

if len(sys.argv) > 4:      # context: the kernels around the N largest gaps
    srt = sorted(rows)
    ends = {}
    big2 = sorted(big, reverse=True)[: int(sys.argv[4])]
    for g in big2:
        # locate by gap size: find consecutive pair
        cur = srt[0][1]
        for i, (s_, e_, n_) in enumerate(srt):
            if s_ - cur == g[0]:
                print("---- gap %.1f us at +%.2f ms" % (g[0] / 1e3, (s_ - srt[0][0]) / 1e6))
                for k in range(max(0, i - 6), min(len(srt), i + 5)):
                    print("   %s %10.3f ms  dur %8.1f us  %s" % (">>" if k == i else "  ", (srt[k][0] - srt[0][0]) / 1e6,
                                                                  (srt[k][1] - srt[k][0]) / 1e3, srt[k][2]))
                break
            cur = max(cur, e_)
