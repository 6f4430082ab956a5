#!/bin/bash
# PMC counters of one three-pass shape: bash tools/s3_pmc.sh <tag> <probe args...>
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")
: > $OUT/${TAG}_pmc.txt
for P in "${PASSES[@]}"; do
  rm -rf /tmp/rpp && timeout 300 rocprofv3 --pmc $P --output-format csv -d /tmp/rpp -o p -- python $GRAFT_REPO_ROOT/tools/s3_probe.py "$@" > $OUT/${TAG}_pmc_pass.log 2>&1
  python - "$OUT/${TAG}_pmc.txt" <<'PY'
import csv, glob, sys, collections
dst = sys.argv[1]
f = glob.glob("/tmp/rpp/**/*counter_collection.csv", recursive=True)
if not f:
    open(dst, "a").write("no counter file\n"); sys.exit(0)
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    k = (r["Kernel_Name"][:110], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")))
    a = agg.setdefault(k, collections.OrderedDict())
    c = a.setdefault(r["Counter_Name"], [0, 0.0])
    c[0] += 1; c[1] += float(r["Counter_Value"])
with open(dst, "a") as fo:
    for k, cs in agg.items():
        if "gemm8" not in k[0]:
            continue
        fo.write("%s grid=%s lds=%s vgpr=%s\n" % k)
        for n, (cnt, v) in cs.items():
            fo.write("    %-28s %14.0f per dispatch (%d dispatches)\n" % (n, v / cnt, cnt))
PY
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc.txt
