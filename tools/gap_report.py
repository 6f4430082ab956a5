"""usage: python tools/gap_report.py <kernel_trace.csv>: splits the trace into graph replays at fill_f32_kernel (the
timestep fill that precedes every gl_unet_forward) and reports, for the LAST replay of each variant, the wall span, the
sum of kernel durations and the idle time between consecutive kernels, plus the per-kernel-name totals."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
seg, cur = [], []
for s, e, n in rows:
    if "fill_f32_kernel" in n:
        if cur:
            seg.append(cur)
        cur = []
    cur.append((s, e, n))
if cur:
    seg.append(cur)
seg = [g for g in seg if len(g) > 300]
for g in (seg[3], seg[-1]) if len(seg) >= 8 else seg[-2:]:
    span = g[-1][1] - g[0][0]
    busy = sum(e - s for s, e, _ in g)
    gaps = [max(0, g[i + 1][0] - g[i][1]) for i in range(len(g) - 1)]
    by = collections.Counter()
    for s, e, n in g:
        k = n.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
        by[k] += e - s
    print(f"replay: {len(g)} kernels, span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle between kernels {sum(gaps) / 1e3:.1f} us "
          f"(mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us)")
    for k, v in by.most_common(16):
        print(f"    {v / 1e3:9.1f} us  {k}")
