#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_strict.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
timeout 600 python tools/engine_ab_probe.py 4 2>&1 | tail -4
FUZZ_STRICT=1 timeout 400 python tools/fuzz_kernels.py 120 6 > $OUT/r6_final_fuzz_kernels_strict.txt 2>&1; tail -n 3 $OUT/r6_final_fuzz_kernels_strict.txt
timeout 900 python bench.py --no-cpu-baseline --steps 3 > $OUT/r6j_bench.json 2> $OUT/r6j_bench.err
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r6j_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "unet_step_ms", d["unet_step_ms"], "strict", d["strict_mode"]["images_per_s"], d["strict_mode"]["unet_forward_ms"])
PY
