#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_strict.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_boundary.py tests/test_gpu_dist.py -x -q -k "strict" 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 1200 python bench.py --no-cpu-baseline --steps 2 > $OUT/r6i_bench.json 2> $OUT/r6i_bench.err
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r6i_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "unet_step_ms", d["unet_step_ms"], "strict", d["strict_mode"]["images_per_s"], d["strict_mode"]["unet_forward_ms"], "line bytes", len(json.dumps(d)))
PY
