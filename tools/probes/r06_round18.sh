# round 6, eighteenth call: the torchrun launch line of the contract at world size 1 (RCCL broadcast / gather legs) for the headline and for Whisper; a second pass of the GPU suite on another box
set -x
mkdir -p gpurun_out/r06r
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r06r/bench_torchrun_n1.json 2> gpurun_out/r06r/bench_torchrun_n1.err; tail -c 400 gpurun_out/r06r/bench_torchrun_n1.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --workload whisper --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06r/bench_torchrun_whisper_n1.json 2> gpurun_out/r06r/bench_torchrun_whisper_n1.err; tail -c 300 gpurun_out/r06r/bench_torchrun_whisper_n1.json
python -m pytest tests -m gpu -q -x > gpurun_out/r06r/pytest_all_second.txt 2>&1; tail -4 gpurun_out/r06r/pytest_all_second.txt
