mkdir -p gpurun_out/r4ae
for c in 0.60 0.66 0.72 0.78 0.86 1.00; do
  for spec in "32 10 20" "64 10 20" "128 10 20" "8 22 42" "16 22 42" "32 22 42"; do
    echo -n "cring=$c $spec: "; timeout -k 10 120 python tools/shard_step.py $spec 20 --debug gemm_ring_cost_permille=$(python3 -c "print(int(round($c*1000)))") 2>/dev/null | tail -1
  done
done > gpurun_out/r4ae/cring.txt 2>&1
cat gpurun_out/r4ae/cring.txt
