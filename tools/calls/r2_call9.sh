#!/bin/bash
# Round 2, call 9: C4 at FULL depth (Llama-3.3-70B, 80 layers, 141 GB bf16, consume_state_dict load) through generate();
# PMC passes on the merged-phase production GEMM next to the four-phase kernel; a bench run on the final tile table.
R=$PWD; OUT=$R/gpurun_out/r2c9; mkdir -p $OUT; export PYTHONPATH=$R
run() { name=$1; shift; echo "== $name"; timeout "$@" > $OUT/$name.log 2>&1; echo "rc=$? ($name)"; tail -${TAIL:-12} $OUT/$name.log; }
TAIL=8 run c4_full 900 python tools/gpu_decode_probe.py 1 32 meta-llama/Llama-3.3-70B-Instruct
TAIL=8 run c4_full_b8 900 python tools/gpu_decode_probe.py 8 32 meta-llama/Llama-3.3-70B-Instruct
OUTSAVE=$OUT; TAIL=40 run pmc 400 bash tools/pmc_gemm_prod.sh
TAIL=2 run bench 300 python bench.py --steps 10 --warmup 3 --gemm-table $OUT/gemm_table.txt
head -16 $OUT/gemm_table.txt
