cd $GRAFT_REPO_ROOT
O=gpurun_out
(python -m pytest tests -m gpu -q 2>&1 | tail -4) > $O/r3_final_gpu_tests.log 2>&1
python bench.py > $O/r3_final_bench.json 2> $O/r3_final_bench.err
python bench.py --config 3 --steps 2 --no-cpu-baseline > $O/r3_final_bench_c3.json 2> $O/r3_final_bench_c3.err
python bench.py --config 5 --steps 2 --no-cpu-baseline > $O/r3_final_bench_c5.json 2> $O/r3_final_bench_c5.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r3_final_smoke.log 2>&1
bash tools/profile_round.sh r3 > $O/r3_final_profile.log 2>&1
KB=1 python tools/kbench.py > $O/r3_final_kbench.txt 2>&1
