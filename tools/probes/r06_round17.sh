# round 6, seventeenth call: the whole -m gpu suite + smoke on the final tree; default bench line; Paraformer bench line
set -x
mkdir -p gpurun_out/r06q
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r06q/pytest_all.txt 2>&1
tail -14 gpurun_out/r06q/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06q/smoke.txt 2>&1; tail -2 gpurun_out/r06q/smoke.txt
python bench.py > gpurun_out/r06q/bench_default.json 2> gpurun_out/r06q/bench_default.err; tail -c 600 gpurun_out/r06q/bench_default.json
python bench.py --workload paraformer --steps 10 2>/dev/null | tail -1 > gpurun_out/r06q/bench_paraformer.json
