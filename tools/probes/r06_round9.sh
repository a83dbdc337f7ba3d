# round 6, ninth measurement: clocks inside the block kernel's A loop (8192), C loop (4096) and phase D hand-over (16384)
set -x
mkdir -p gpurun_out/r06i
for o in 4096 8192 16384; do ASR_SANM_BLOCK8_OPT=$o ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -23; done > gpurun_out/r06i/inner_clocks.txt 2>&1
cat gpurun_out/r06i/inner_clocks.txt
