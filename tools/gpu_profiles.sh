#!/bin/bash
# dev: the profiles committed under profiles/ for this round (one gpurun call); everything lands in gpurun_out/r03/
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
# 1. the driver's bench command, unprofiled: the line
timeout -k 10 400 python bench.py > $O/bench_line.log 2>&1; tail -1 $O/bench_line.log > $O/r03_bench_line.json; echo "bench rc=$?"
# 2. kernel stats of the same step
rm -rf $O/ks; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o r -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-variants > $O/ks.log 2>&1
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/r03_bench_kernel_stats.csv 2>/dev/null; rm -rf $O/ks
# 3. PMC passes (eager launches, separate passes per counter group)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_'); rm -rf $O/pmc_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-variants > $O/pmc_$n.log 2>&1
  echo "pmc $n rc=$?"
done
for k in clip_adam grad_sqnorm g_times_w gather_pool dw_partial4 prod_gemm_b16 build_g head_fwd_pool; do
  python tools/pmc_summary.py $k $O/r03_${k}_pmc.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum_TCC_MISS_sum > /dev/null 2>&1 || echo "pmc summary $k failed"
done
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum_TCC_MISS_sum
# 4. the bf16 class (BASELINE configs 3 and 5 name it)
timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline --no-variants > $O/bench_bf16.log 2>&1; tail -1 $O/bench_bf16.log > $O/r03_bench_line_bf16.json
# 5. secondary models
for m in narre datt siamese; do
  timeout -k 10 300 python tools/bench_models.py $m > $O/models_$m.log 2>&1; grep '^{' $O/models_$m.log | tail -1 > $O/r03_${m}_bench_line.json
  rm -rf $O/ks_$m; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$m -o r -- python3 tools/bench_models.py $m > $O/ks_$m.log 2>&1
  cp $(find $O/ks_$m -name '*kernel_stats.csv' | head -1) $O/r03_${m}_kernel_stats.csv 2>/dev/null; rm -rf $O/ks_$m
done
timeout -k 10 300 python tools/bench_models.py narre --precision=bf16 > $O/models_narre_bf16.log 2>&1; grep '^{' $O/models_narre_bf16.log | tail -1 > $O/r03_narre_bf16_bench_line.json
ls -la $O | head -40
for f in $O/r03_bench_line.json $O/r03_bench_line_bf16.json $O/r03_narre_bench_line.json $O/r03_narre_bf16_bench_line.json $O/r03_datt_bench_line.json $O/r03_siamese_bench_line.json; do echo "--- $f"; cut -c1-400 $f; done
exit 0
