"""usage: python tools/timeline_diff.py <timeline A> <timeline B>: two tools/timeline_report.py listings of the same launch sequence
(same kernel count) side by side, per-launch difference, and the difference summed by kernel name."""
import collections
import re
import sys


def load(p):
    rows = []
    for ln in open(p):
        m = re.match(r"\s*(\d+)\s+([\d.]+) us\s+blocks\s+(\d+) x(\d+)\s+(.*)", ln)
        if m:
            rows.append((float(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5).strip()))
    return rows


a, b = load(sys.argv[1]), load(sys.argv[2])
print(f"# A: {len(a)} kernels {sum(r[0] for r in a):.1f} us   B: {len(b)} kernels {sum(r[0] for r in b):.1f} us")
by = collections.OrderedDict()
if len(a) == len(b):
    for i, (ra, rb) in enumerate(zip(a, b)):
        d = ra[0] - rb[0]
        if abs(d) >= 1.0:
            print(f"{i:4d} {ra[0]:8.1f} {rb[0]:8.1f} {d:+8.1f}  {ra[3]} [{ra[1]}x{ra[2]}]" + ("" if ra[3] == rb[3] else f"  | B: {rb[3]} [{rb[1]}x{rb[2]}]"))
for tag, rows in (("A", a), ("B", b)):
    for r in rows:
        e = by.setdefault(r[3], {"A": [0, 0.0], "B": [0, 0.0]})
        e[tag][0] += 1
        e[tag][1] += r[0]
print("# by kernel name: launches A / B, us A / B, A - B")
for n, e in sorted(by.items(), key=lambda kv: -(kv[1]["A"][1] - kv[1]["B"][1])):
    print(f"# {e['A'][0]:4d} {e['B'][0]:4d} {e['A'][1]:9.1f} {e['B'][1]:9.1f} {e['A'][1] - e['B'][1]:+9.1f}  {n}")
