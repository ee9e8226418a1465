# Session r6-14: the round's final tree -- GPU suite, smoke, the bench line (driver's arguments), profiles.
mkdir -p gpurun_out/s14
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s14/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/s14/pytest_gpu.log)
grep -a "passed\|failed" gpurun_out/s14/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s14/smoke.log 2>&1; echo "rc $?" >> gpurun_out/s14/smoke.log); tail -n 2 gpurun_out/s14/smoke.log
(timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/s14/bench.json 2> gpurun_out/s14/bench.err; echo "rc $?" >> gpurun_out/s14/bench.err)
grep -a "SUMMARY\|^rc" gpurun_out/s14/bench.err | tail -2 | cut -c1-420
wc -c gpurun_out/s14/bench.json
cp gpurun_out/bench_full.json gpurun_out/s14/bench_full.json 2>/dev/null
(timeout 900 bash tools/profile_round.sh r06b > gpurun_out/s14/profile.log 2>&1; echo "rc $?" >> gpurun_out/s14/profile.log)
tail -n 2 gpurun_out/s14/profile.log
