# round 6, last call: the whole -m gpu suite + smoke on the final tree, then the round's profile set (tools/profile_round.sh)
set -x
mkdir -p gpurun_out/r06x
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r06x/pytest_all.txt 2>&1
tail -9 gpurun_out/r06x/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06x/smoke.txt 2>&1; tail -1 gpurun_out/r06x/smoke.txt
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log
