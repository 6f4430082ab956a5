#!/bin/bash
# Per-kernel PMC counters of the hot shapes (tools/gemm_probe.py), in separate rocprofv3 --pmc passes (no other trace
# domains, as the pool requires).  usage: bash tools/pmc_round.sh <tag> [probe args...]  -> gpurun_out/prof/<tag>_pmc_hot.txt
set -u
TAG=${1:-r2}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE"
        "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE")
: > $OUT/${TAG}_pmc_hot.txt
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rm -rf /tmp/rpp && timeout 600 rocprofv3 --pmc $P --output-format csv -d /tmp/rpp -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py "$@" > $OUT/${TAG}_pmc_pass$i.log 2>&1
  python - "$OUT/${TAG}_pmc_hot.txt" <<'PY'
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
        if "at::native" in k[0] or "transpose_v" in k[0]:
            continue
        fo.write("%s grid=%s lds=%s vgpr=%s\n" % k)
        for n, (cnt, v) in cs.items():
            fo.write("    %-28s %14.0f per dispatch (%d dispatches)\n" % (n, v / cnt, cnt))
PY
done
cat $OUT/${TAG}_pmc_hot.txt
