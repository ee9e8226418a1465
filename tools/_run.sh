cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02 > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err; echo "rc=$?"
tail -2 gpurun_out/r02_bench_final.err
