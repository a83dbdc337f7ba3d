cd /root/repo
for e in "" "ASR_SANM_FUSED=0" "ASR_SKINNY_M144=1" "ASR_SANM_FUSED=0 ASR_SKINNY_M144=1" "ASR_GEMM_T144=0" "ASR_LN_FUSED=0" "ASR_GEMM_SPLITK=0" "ASR_SANM_BLOCK_MIN=1"; do
  echo "== $e: $(env $e python tools/probes/bf16_b1_trace.py 2>&1 | grep 'ms per chunk')"
done
