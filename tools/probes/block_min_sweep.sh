# batches of 1 .. 48 windows through the 8-wave block kernel (ASR_SANM_BLOCK_MIN=1) and through the four launches per block (=99): where the crossover is
for b in 1 2 4 8 16 32 48; do for m in 1 99; do
  r=$(ASR_SANM_BLOCK_MIN=$m timeout 300 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "batch $b block_min $m ms_per_step $r"
done; done
