#!/bin/bash
# dev: the profiles committed under profiles/ for round 4 (one gpurun call); everything lands in gpurun_out/r04/
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
# 1. the driver's bench command, unprofiled: the line (with the `configs` object and the CPU baseline)
timeout -k 10 500 python bench.py > $O/bench_line.log 2>&1; tail -1 $O/bench_line.log > $O/r04_bench_line.json; echo "bench rc=$?"
# 2. kernel stats of the headline step (no configs: their kernels share names with the headline's)
rm -rf $O/ks; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o r -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-variants --no-configs > $O/ks.log 2>&1
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/r04_bench_kernel_stats.csv 2>/dev/null; rm -rf $O/ks
# 3. PMC passes of the headline step (eager launches, separate passes per counter group)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  n=$(echo $c | tr ' ' '_'); rm -rf $O/pmc_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-variants --no-configs > $O/pmc_$n.log 2>&1
  echo "pmc $n rc=$?"
done
SQ=$O/pmc_GRBM_GUI_ACTIVE_SQ_BUSY_CYCLES_SQ_VALU_MFMA_BUSY_CYCLES_SQ_WAVE_CYCLES
for k in clip_adam grad_sqnorm g_times_w gather_pool dw_partial4 prod_gemm_b16d build_g head_fwd_pool; do
  python tools/pmc_summary.py $k $O/r04_${k}_pmc.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum_TCC_MISS_sum $SQ > /dev/null 2>&1 || echo "pmc summary $k failed"
done
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum_TCC_MISS_sum $SQ
# 4. the bf16 class
timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline --no-variants --no-configs > $O/bench_bf16.log 2>&1; tail -1 $O/bench_bf16.log > $O/r04_bench_line_bf16.json
# 5. secondary models: lines (bench_models.py) and clean per-step kernel stats (graph replays only: dev_count_launches.py)
for m in narre datt siamese; do
  timeout -k 10 300 python tools/bench_models.py $m > $O/models_$m.log 2>&1; grep '^{' $O/models_$m.log | tail -1 > $O/r04_${m}_bench_line.json
done
timeout -k 10 300 python tools/bench_models.py narre --precision=bf16 > $O/models_narre_bf16.log 2>&1; grep '^{' $O/models_narre_bf16.log | tail -1 > $O/r04_narre_bf16_bench_line.json
for m in narre datt; do
  rm -rf $O/ks_$m; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$m -o r -- python3 tools/dev_count_launches.py $m 60 > $O/ks_$m.log 2>&1
  cp $(find $O/ks_$m -name '*kernel_stats.csv' | head -1) $O/r04_${m}_kernel_stats.csv 2>/dev/null
  f=$(find $O/ks_$m -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python tools/step_timeline.py $f > $O/r04_${m}_step_timeline.txt 2>/dev/null
  rm -rf $O/ks_$m
done
# 6. D-ATT: HBM / L2 counters of its long kernels (eager step)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  n=$(echo $c | tr ' ' '_'); rm -rf $O/dpmc_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/dpmc_$n -o r -- python3 tools/bench_models.py datt --no-graph > $O/dpmc_$n.log 2>&1
done
for k in gather_pool g_times_w prod_gemm_b16k_kernel gg_rows global_gate_fwd; do
  python tools/pmc_summary.py $k $O/r04_datt_${k}_pmc.json $O/dpmc_FETCH_SIZE $O/dpmc_WRITE_SIZE $O/dpmc_TCC_HIT_sum_TCC_MISS_sum $O/dpmc_GRBM_GUI_ACTIVE_SQ_BUSY_CYCLES_SQ_VALU_MFMA_BUSY_CYCLES_SQ_WAVE_CYCLES > /dev/null 2>&1 || echo "datt pmc summary $k failed"
done
rm -rf $O/dpmc_FETCH_SIZE $O/dpmc_WRITE_SIZE $O/dpmc_TCC_HIT_sum_TCC_MISS_sum $O/dpmc_GRBM_GUI_ACTIVE_SQ_BUSY_CYCLES_SQ_VALU_MFMA_BUSY_CYCLES_SQ_WAVE_CYCLES
ls -la $O | head -50
for f in $O/r04_bench_line_bf16.json $O/r04_narre_bench_line.json $O/r04_narre_bf16_bench_line.json $O/r04_datt_bench_line.json $O/r04_siamese_bench_line.json; do echo "--- $f"; cut -c1-300 $f; done
exit 0
