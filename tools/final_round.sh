#!/bin/bash
# Round-end evidence on the GPU box (run via gpurun): bash tools/final_round.sh <tag>   e.g. r4
# GPU suite, the three bench configs, smoke, rocprofv3 kernel stats + PMC traffic, per-shape table, fuzz sweeps.
T=${1:-r5}
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
(python -m pytest tests -m gpu -q 2>&1 | tail -4) > $O/${T}_final_gpu_tests.log 2>&1
python bench.py > $O/${T}_final_bench.json 2> $O/${T}_final_bench.err
python bench.py --config 3 --steps 2 --no-cpu-baseline > $O/${T}_final_bench_c3.json 2> $O/${T}_final_bench_c3.err
python bench.py --config 5 --steps 2 --no-cpu-baseline > $O/${T}_final_bench_c5.json 2> $O/${T}_final_bench_c5.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${T}_final_smoke.log 2>&1
bash tools/profile_round.sh $T > $O/${T}_final_profile.log 2>&1
KB=1 python tools/kbench.py > $O/${T}_final_kbench.txt 2>&1
python tools/fuzz_kernels.py 150 4 > $O/${T}_final_fuzz_kernels.txt 2>&1
python tools/fuzz_engine.py 150 4 > $O/${T}_final_fuzz_engine.txt 2>&1
tail -2 $O/${T}_final_gpu_tests.log; cut -c1-300 $O/${T}_final_bench.json; tail -1 $O/${T}_final_fuzz_kernels.txt; tail -1 $O/${T}_final_fuzz_engine.txt
