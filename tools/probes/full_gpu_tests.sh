set -x
mkdir -p gpurun_out/full
python -m pytest tests -m gpu -q > gpurun_out/full/pytest_all.txt 2>&1
tail -15 gpurun_out/full/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.txt 2>&1
tail -3 gpurun_out/full/smoke.txt
