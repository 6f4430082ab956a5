#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -k "split_attention" 2>&1 | tail -3
for i in 1 2; do python tools/attn_split_probe.py; KB_OPTS=53=2 python tools/attn_split_probe.py; done
python tools/attn_split_probe.py 40 4096 4126 8 10
python tools/attn_split_probe.py 40 4096 77 8 20
KB_OPTS=53=2 python tools/attn_split_probe.py 40 4096 77 8 20
python tools/attn_split_probe.py 40 9216 9246 4 5
KB_OPTS=53=2 python tools/attn_split_probe.py 40 9216 9246 4 5
