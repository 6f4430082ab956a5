#!/bin/bash
# Per-kernel timelines of one fuser-off and one fuser-on graph replay (tools/gap_probe.py under rocprofv3 --kernel-trace) for the default
# options and for an A/B variant given as gl_set_option pairs:   bash tools/timeline_ab.sh <tag> <key=value> [key=value ...]
# Writes gpurun_out/<tag>_timeline_{default,variant}_{off,on}.txt and a per-launch diff of the fuser-off replays.
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in default variant; do
  ARGS=""; [ $V = variant ] && ARGS="$*"
  rm -rf /tmp/tl_$V && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$V -o t -- python $GRAFT_REPO_ROOT/tools/gap_probe.py 4 $ARGS > /tmp/tl_$V.log 2>&1
  F=$(find /tmp/tl_$V -name "*kernel_trace.csv" | head -1)
  for W in off on; do python $GRAFT_REPO_ROOT/tools/timeline_report.py $F $W > $OUT/${TAG}_timeline_${V}_$W.txt; done
done
python $GRAFT_REPO_ROOT/tools/timeline_diff.py $OUT/${TAG}_timeline_default_off.txt $OUT/${TAG}_timeline_variant_off.txt > $OUT/${TAG}_timeline_diff_off.txt
tail -40 $OUT/${TAG}_timeline_diff_off.txt
