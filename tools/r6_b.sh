#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
GLIGEN_HIP_LIB=libgligen_hip_ph2.so timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -k "three_pass or conv3x3 or geglu" 2>&1 | tail -5 > $OUT/r6b_ph2_tests.log
tail -2 $OUT/r6b_ph2_tests.log
KB_STRICT=1 timeout 600 python tools/kbench.py gemm conv > $OUT/r6b_kb_strict_ph3.txt 2>&1
GLIGEN_HIP_LIB=libgligen_hip_ph2.so KB_STRICT=1 timeout 600 python tools/kbench.py gemm conv > $OUT/r6b_kb_strict_ph2.txt 2>&1
KB_STRICT=1 timeout 600 python tools/kbench.py attn > $OUT/r6b_kb_attn_v0.txt 2>&1
KB_STRICT=1 KB_OPTS=53=1 timeout 600 python tools/kbench.py attn > $OUT/r6b_kb_attn_v1.txt 2>&1
grep -i "total" $OUT/r6b_kb_*.txt
cat $OUT/r6b_kb_attn_v0.txt $OUT/r6b_kb_attn_v1.txt
