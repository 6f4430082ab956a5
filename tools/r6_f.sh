#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for o in "46=11" "46=0" "46=1" "46=2" "46=11,34=8" "46=11,34=16"; do
  KB_STRICT=1 KB_OPTS=$o timeout 600 python tools/kbench.py gemm conv > "$OUT/r6f_kb_strict_$o.txt" 2>&1
  echo "== $o"; grep -i total "$OUT/r6f_kb_strict_$o.txt"
done
