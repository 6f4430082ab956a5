#!/bin/bash
# round 6: the dedicated three-pass loop (key 52) -- correctness, per-shape A/B against the K-walk, strict-mode profile
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/prof
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -s 2>&1 | tail -60 > $OUT/r6a_strict_tests.log
tail -3 $OUT/r6a_strict_tests.log
KB_STRICT=1 timeout 600 python tools/kbench.py gemm conv > $OUT/r6a_kb_strict_s3.txt 2>&1
KB_STRICT=1 KB_OPTS=52=0 timeout 600 python tools/kbench.py gemm conv > $OUT/r6a_kb_strict_kwalk.txt 2>&1
grep -i "total" $OUT/r6a_kb_strict_s3.txt $OUT/r6a_kb_strict_kwalk.txt
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
rm -rf /tmp/rp1 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-strict --strict-main --no-hot-kernel --opt 52=$v > $OUT/prof/r6a_strict_s3_${v}_trace.log 2>&1
find /tmp/rp1 -name "*kernel_stats.csv" -exec cp {} $OUT/prof/r6a_strict_s3_${v}_kernel_stats.csv \;
tail -1 $OUT/prof/r6a_strict_s3_${v}_trace.log | cut -c1-400
done
