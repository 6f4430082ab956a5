#!/bin/bash
# Copies what tools/final_round.sh left under gpurun_out/ into profiles/ (the tracked, judged evidence) for round tag $1 (default r6).
set -eu
T=${1:-r6}
cd "$(dirname "$0")/.."
H="$(git rev-parse --short HEAD)$(git diff --quiet || echo +dirty)"
for M in "" "_strict"; do
  cp gpurun_out/prof/${T}${M}_kernel_stats.csv gpurun_out/prof/${T}${M}_pmc_FETCH_SIZE.csv gpurun_out/prof/${T}${M}_pmc_WRITE_SIZE.csv gpurun_out/prof/${T}${M}_bench_under_trace.log profiles/
  python tools/traffic_from_pmc.py profiles/${T}${M}_pmc_FETCH_SIZE.csv profiles/${T}${M}_pmc_WRITE_SIZE.csv profiles/${T}${M}_traffic.json 13 "$H" > /dev/null
  python tools/stats_breakdown.py profiles/${T}${M}_kernel_stats.csv > profiles/${T}${M}_kernel_classes.txt
done
cp gpurun_out/${T}_final_bench.json profiles/${T}_bench.json
cp gpurun_out/${T}_final_bench_c3.json profiles/${T}_bench_c3.json
cp gpurun_out/${T}_final_bench_c5.json profiles/${T}_bench_c5.json
cp gpurun_out/${T}_final_bench_strict_main.json profiles/${T}_bench_strict_main.json
grep -v amdgpu.ids gpurun_out/${T}_final_kbench.txt > profiles/${T}_kbench_shapes.txt
( echo "# strict operand forms ([hi | lo] activations, [Whi | Wlo] weights, three passes, split attention): round-6 kernels (three-pass main loop, key 52 = 1;"
  echo "# software-pipelined split attention, key 53 = 0), then the round-5 forms on the same box (K-walk, key 52 = 0; attn_split_kernel, key 53 = 2)"
  grep -v amdgpu.ids gpurun_out/${T}_final_kbench_strict.txt
  echo; echo "# ---- round-5 forms (KB_OPTS=52=0,53=2), same box, same run"
  grep -v amdgpu.ids gpurun_out/${T}_final_kbench_strict_r5forms.txt ) > profiles/${T}_kbench_shapes_strict.txt
( echo "# split-fp16 attention, d = 40, 4096 x 4096, 2B = 8: PMC passes (tools/attn_pmc.sh): attn_split_pipe_kernel (round 6), then attn_split_kernel (round 5)"
  cat gpurun_out/prof/${T}_pipe_pmc_attn.txt; python tools/pmc_summary.py gpurun_out/prof/${T}_pipe_pmc_attn.txt
  cat gpurun_out/prof/${T}_r5kernel_pmc_attn.txt; python tools/pmc_summary.py gpurun_out/prof/${T}_r5kernel_pmc_attn.txt ) > profiles/${T}_pmc_attn_split.txt
cp gpurun_out/${T}_final_gpu_tests.log profiles/${T}_gpu_tests.log
(tail -3 gpurun_out/${T}_final_fuzz_kernels.txt) > profiles/${T}_fuzz_kernels.txt 2>/dev/null || true
(tail -3 gpurun_out/${T}_final_fuzz_engine.txt) > profiles/${T}_fuzz_engine.txt 2>/dev/null || true
(tail -3 gpurun_out/${T}_final_fuzz_engine_strict.txt) > profiles/${T}_fuzz_engine_strict.txt 2>/dev/null || true
python - "$T" <<'PY'
import json, sys
t = sys.argv[1]
for f in ("bench", "bench_c3", "bench_c5", "bench_strict_main"):
    d = json.loads(open(f"profiles/{t}_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["unit"], "ms/step", d["ms_per_step"], "unet_step_ms", d["unet_step_ms"], "frac", d["roofline"]["frac"], "strict", (d.get("strict_mode") or {}).get("images_per_s"))
PY
tail -2 profiles/${T}_gpu_tests.log
