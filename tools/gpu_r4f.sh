#!/bin/bash
# dev (round 4): SQ counters of D-ATT's product-table GEMM, rows-stationary form against the ring form
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4f; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1 || true
grep -o "SQ_[A-Z_0-9]*" $O/avail.txt | sort -u | tr '\n' ' ' | cut -c1-3000
echo
for f in 1 0; do
  export RBR_GEMM_ROWS_STATIONARY=$f
  i=0
  for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p${f}_$i -o r -- python3 tools/bench_models.py datt --no-graph > $O/p${f}_$i.log 2>&1
    echo "stationary=$f pass $i rc=$?"
    python tools/pmc_counters.py prod_gemm_b16 $O/p${f}_$i
    rm -rf $O/p${f}_$i
  done
done
