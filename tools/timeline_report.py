"""usage: python tools/timeline_report.py <kernel_trace.csv> [on|off]: lists every kernel of the LAST graph replay of the
fuser-on / fuser-off forward (tools/gap_probe.py under rocprofv3 --kernel-trace) in launch order with its grid and
duration, then sums the replay by duration class -- where the time of the ~520 kernel boundaries actually sits."""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1),
                 int(r.get("Grid_Size_Z", 1) or 1)))
rows.sort()
seg, cur = [], []
for row in rows:
    if "fill_f32_kernel" in row[2]:
        if cur:
            seg.append(cur)
        cur = []
    cur.append(row)
if cur:
    seg.append(cur)
seg = [g for g in seg if len(g) > 300]
which = sys.argv[2] if len(sys.argv) > 2 else "off"
g = seg[-1] if which == "off" else seg[len(seg) // 2 - 1]


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)(I|E)", n)
    if m:
        return m.group(1)
    return n.split("(")[0][:48]


print(f"# {len(g)} kernels, span {(g[-1][1] - g[0][0]) / 1e3:.1f} us")
classes = [(5, "< 5 us"), (10, "5-10 us"), (20, "10-20 us"), (50, "20-50 us"), (1e9, ">= 50 us")]
acc = {c: [0, 0.0] for _, c in classes}
for i, (s, e, n, gx, wx, gz) in enumerate(g):
    d = (e - s) / 1e3
    for lim, c in classes:
        if d < lim:
            acc[c][0] += 1
            acc[c][1] += d
            break
    print(f"{i:4d} {d:8.1f} us  blocks {gx // max(wx, 1):6d} x{gz:<3d} {short(n)}")
print("# by duration class:")
for _, c in classes:
    print(f"#   {c:9s} {acc[c][0]:4d} kernels {acc[c][1]:9.1f} us")
