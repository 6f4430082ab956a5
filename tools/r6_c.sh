#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -s -k "split_attention" 2>&1 | tail -30 > $OUT/r6c_attn_tests.log
cat $OUT/r6c_attn_tests.log | grep -E "split attention|passed|failed|Error|error" | head -40
KB_STRICT=1 timeout 600 python tools/kbench.py attn > $OUT/r6c_kb_attn_pipe.txt 2>&1
KB_STRICT=1 KB_OPTS=53=2 timeout 600 python tools/kbench.py attn > $OUT/r6c_kb_attn_old.txt 2>&1
cat $OUT/r6c_kb_attn_pipe.txt $OUT/r6c_kb_attn_old.txt
