#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
AB_PARITY=1 timeout 900 python tools/engine_ab_probe.py 4 pure16:38=0,41=0,42=0,45=0 no45:45=0 no41_45:41=0,45=0 no42:42=0 > $OUT/r6e_ab_default_precision.txt 2>&1
tail -16 $OUT/r6e_ab_default_precision.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -s -k "config0_strict" 2>&1 | grep -E "STRICT|passed|failed" > $OUT/r6e_strict_sampler.log
cat $OUT/r6e_strict_sampler.log
timeout 1200 python bench.py --no-cpu-config1 > $OUT/r6e_bench.json 2> $OUT/r6e_bench.err
tail -c 3000 $OUT/r6e_bench.json
