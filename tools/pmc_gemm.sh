#!/bin/bash
# PMC passes for the GEMM kernel variants (separate passes: SQ has 8 slots, TCC 4)
R=$PWD; OUT=$R/gpurun_out/pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
rocprofv3 -L > $OUT/counters.txt 2>&1
for V in 4 8; do
  timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/sq_v$V -o p --output-format csv -- timeout 300 python $R/tools/gpu_gemm_pmc.py $V 2528 28672 4096 > $OUT/sq_v$V.log 2>&1
  timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/lds_v$V -o p --output-format csv -- timeout 300 python $R/tools/gpu_gemm_pmc.py $V 2528 28672 4096 > $OUT/lds_v$V.log 2>&1
  timeout 420 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/tcc_v$V -o p --output-format csv -- timeout 300 python $R/tools/gpu_gemm_pmc.py $V 2528 28672 4096 > $OUT/tcc_v$V.log 2>&1
  timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_v$V -o p --output-format csv -- timeout 300 python $R/tools/gpu_gemm_pmc.py $V 2528 28672 4096 > $OUT/fetch_v$V.log 2>&1
done
find $OUT -name "*.csv" | head -40
