#!/bin/bash
# Round-end evidence on the GPU box (run via gpurun): bash tools/final_round.sh <tag>   e.g. r6
# GPU suite, the three bench configs, smoke, rocprofv3 kernel stats + PMC traffic (default and strict mode), per-shape tables (default, strict
# three-pass loop, strict K-walk), split-attention PMC, fuzz sweeps (default and strict).
T=${1:-r6}
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
(python -m pytest tests -m gpu -q 2>&1 | tail -4) > $O/${T}_final_gpu_tests.log 2>&1
python bench.py > $O/${T}_final_bench.json 2> $O/${T}_final_bench.err
python bench.py --config 3 --steps 2 --no-cpu-baseline > $O/${T}_final_bench_c3.json 2> $O/${T}_final_bench_c3.err
python bench.py --config 5 --steps 2 --no-cpu-baseline > $O/${T}_final_bench_c5.json 2> $O/${T}_final_bench_c5.err
python bench.py --strict-main --steps 2 --no-cpu-baseline > $O/${T}_final_bench_strict_main.json 2> $O/${T}_final_bench_strict_main.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${T}_final_smoke.log 2>&1
bash tools/profile_round.sh $T > $O/${T}_final_profile.log 2>&1
bash tools/profile_round.sh $T strict > $O/${T}_final_profile_strict.log 2>&1
KB=1 python tools/kbench.py > $O/${T}_final_kbench.txt 2>&1
KB_STRICT=1 python tools/kbench.py gemm conv attn > $O/${T}_final_kbench_strict.txt 2>&1
KB_STRICT=1 KB_OPTS=52=0,53=2 python tools/kbench.py gemm conv attn > $O/${T}_final_kbench_strict_r5forms.txt 2>&1
bash tools/attn_pmc.sh ${T}_pipe > /dev/null 2>&1
bash tools/attn_pmc.sh ${T}_r5kernel 53=2 > /dev/null 2>&1
python tools/fuzz_kernels.py 150 4 > $O/${T}_final_fuzz_kernels.txt 2>&1
python tools/fuzz_engine.py 150 4 > $O/${T}_final_fuzz_engine.txt 2>&1
FUZZ_STRICT=1 python tools/fuzz_engine.py 100 4 > $O/${T}_final_fuzz_engine_strict.txt 2>&1
tail -2 $O/${T}_final_gpu_tests.log; cut -c1-300 $O/${T}_final_bench.json; tail -1 $O/${T}_final_fuzz_kernels.txt; tail -1 $O/${T}_final_fuzz_engine.txt; tail -1 $O/${T}_final_fuzz_engine_strict.txt
