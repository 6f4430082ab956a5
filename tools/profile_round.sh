#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run via gpurun):
#   1. --kernel-trace --stats of the default bench command          -> gpurun_out/prof/<tag>_kernel_stats.csv
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes  -> gpurun_out/prof/<tag>_pmc_{fetch,write}.csv
# (PMC passes are kept apart from trace domains other than kernel-trace, as the pool requires.)
set -u
TAG=${1:-r1}
# second argument "strict": the same passes with the MAIN loop of bench.py in strict mode (--strict-main), files <tag>_strict_*
MODE=${2:-default}
EXTRA=""
if [ "$MODE" = "strict" ]; then EXTRA="--strict-main"; TAG=${TAG}_strict; fi
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-strict --no-hot-kernel $EXTRA"
rm -rf /tmp/rp1 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -o $TAG -- $BENCH > $OUT/${TAG}_bench_under_trace.log 2>&1
find /tmp/rp1 -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp2 && timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/rp2 -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --plms-steps 10 --no-cpu-baseline --no-strict --no-vae --no-hot-kernel $EXTRA > $OUT/${TAG}_bench_under_pmc_$c.log 2>&1
  python - "$c" "$OUT/${TAG}_pmc_$c.csv" <<'PY'
import csv, glob, sys, collections
name, dst = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/rp2/**/*counter_collection.csv", recursive=True)
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != name:
        continue
    k = r["Kernel_Name"]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
with open(dst, "w") as fo:
    fo.write("kernel,dispatches,sum_%s\n" % name)
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fo.write('"%s",%d,%.1f\n' % (k.replace('"', "'"), n, v))
print(name, "kernels", len(agg), "total", sum(v for _, v in agg.values()))
PY
done
tail -1 $OUT/${TAG}_bench_under_trace.log | cut -c1-300
head -12 $OUT/${TAG}_kernel_stats.csv
