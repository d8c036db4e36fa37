"""Who runs alone: python tools/trace_solo.py <kernel_trace.csv> [window_ms]
For the last <window_ms> of a rocprofv3 kernel trace (one replayed step), the time each kernel name spends as the ONLY kernel on the GPU,
split by whether its grid fills the chip (workgroups >= 256 CUs x 2), and the time spent with 2 / 3+ kernels resident."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid))
tend = max(r[1] for r in rows)
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 99e6
rows = [r for r in rows if r[0] >= tend - win]
ev = []
for i, (s, e, n, g) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, 0, i))
ev.sort()
active = set(); last = ev[0][0]
solo = defaultdict(float); solo_small = defaultdict(float); total = defaultdict(float); conc = defaultdict(float); launches = defaultdict(int)
for t, kind, i in ev:
    dt = t - last
    if dt > 0 and active:
        conc[min(len(active), 3)] += dt
        if len(active) == 1:
            j = next(iter(active)); n = rows[j][2]
            solo[n] += dt
            if rows[j][3] < 512:
                solo_small[n] += dt
    last = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
for s, e, n, g in rows:
    total[n] += e - s; launches[n] += 1
print("resident kernels: 1: %.1f ms  2: %.1f ms  3+: %.1f ms" % (conc[1] / 1e6, conc[2] / 1e6, conc[3] / 1e6))
print("%-86s %6s %8s %8s %8s" % ("kernel", "n", "total", "solo", "solo<512wg"))
for n in sorted(solo, key=lambda k: -solo[k])[:45]:
    print("%-86s %6d %8.2f %8.2f %8.2f" % (n[:86], launches[n], total[n] / 1e6, solo[n] / 1e6, solo_small[n] / 1e6))
print("solo with small grids, all kernels: %.2f ms" % (sum(solo_small.values()) / 1e6))
