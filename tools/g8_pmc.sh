#!/bin/bash
# PMC counters of the 8-wave GEMM / conv kernel on one probe case (separate rocprofv3 --pmc passes, no trace domains).
# usage: bash tools/g8_pmc.sh <tag> <case substring>   -> gpurun_out/prof/<tag>_pmc_g8.txt
set -u
TAG=${1:-r3}; ONLY=${2:-"conv 8x64^2 320->320"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE"
        "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"
        "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"
        "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
        "TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCP_RFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE")
: > $OUT/${TAG}_pmc_g8.txt
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rm -rf /tmp/rpp && G8_ONLY="$ONLY" timeout 300 rocprofv3 --pmc $P --output-format csv -d /tmp/rpp -o p -- python $GRAFT_REPO_ROOT/tools/g8_probe.py spin > $OUT/${TAG}_pmc_g8_pass$i.log 2>&1
  python - "$OUT/${TAG}_pmc_g8.txt" <<'PY'
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
cat $OUT/${TAG}_pmc_g8.txt
