#!/bin/bash
# GPU probe: grid shapes of the decode GEMM (ASR_DECODE_NT / ASR_DECODE_KS pin them) on the cold-weight launch chains of the Whisper decoder shapes
echo "== plan"; python tools/probes/decode_gemm_chain2.py ${1:-32,64} 2>&1 | grep "^M="
for nt in 1 2; do for ks in 1 2 3 4 5 8; do echo "== NT=$nt KS=$ks"; ASR_DECODE_NT=$nt ASR_DECODE_KS=$ks python tools/probes/decode_gemm_chain2.py ${1:-32,64} 2>&1 | grep "^M=" | grep -v "six"; done; done
