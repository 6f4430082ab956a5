#!/bin/bash
# Copies what tools/final_round.sh left under gpurun_out/ into profiles/ (the tracked, judged evidence) for round tag $1 (default r3).
set -eu
T=${1:-r5}
cd "$(dirname "$0")/.."
cp gpurun_out/prof/${T}_kernel_stats.csv gpurun_out/prof/${T}_pmc_FETCH_SIZE.csv gpurun_out/prof/${T}_pmc_WRITE_SIZE.csv gpurun_out/prof/${T}_bench_under_trace.log profiles/
cp gpurun_out/${T}_final_bench.json profiles/${T}_bench.json
cp gpurun_out/${T}_final_bench_c3.json profiles/${T}_bench_c3.json
cp gpurun_out/${T}_final_bench_c5.json profiles/${T}_bench_c5.json
grep -v amdgpu.ids gpurun_out/${T}_final_kbench.txt > profiles/${T}_kbench_shapes.txt
cp gpurun_out/${T}_final_gpu_tests.log profiles/${T}_gpu_tests.log
(tail -3 gpurun_out/${T}_final_fuzz_kernels.txt) > profiles/${T}_fuzz_kernels.txt 2>/dev/null || true
(tail -3 gpurun_out/${T}_final_fuzz_engine.txt) > profiles/${T}_fuzz_engine.txt 2>/dev/null || true
python tools/traffic_from_pmc.py profiles/${T}_pmc_FETCH_SIZE.csv profiles/${T}_pmc_WRITE_SIZE.csv profiles/${T}_traffic.json 13 "$(git rev-parse --short HEAD)$(git diff --quiet || echo +dirty)" > /dev/null
python tools/stats_breakdown.py profiles/${T}_kernel_stats.csv > profiles/${T}_kernel_classes.txt
python - "$T" <<'PY'
import json, sys
t = sys.argv[1]
for f in ("bench", "bench_c3", "bench_c5"):
    d = json.loads(open(f"profiles/{t}_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["unit"], "ms/step", d["ms_per_step"], "unet_step_ms", d["unet_step_ms"], "frac", d["roofline"]["frac"])
PY
tail -2 profiles/${T}_gpu_tests.log
