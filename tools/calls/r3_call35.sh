#!/bin/bash
# round 3, call 35: streamed weight transposes: bit-identity test, cost on C2 (off / on, same box), and Llama-3.3-70B training on one GPU
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "streamed or two_stream" 2>&1 | grep -E "passed|failed|rror" | tail -3
run() { timeout 900 python bench.py --steps $3 --warmup 2 --no-cpu-baseline $2 > gpurun_out/r3c35_$1.json 2> gpurun_out/r3c35_$1.err
  python -c "
import json; d=json.load(open('gpurun_out/r3c35_$1.json')); print('%-14s ms/step %.2f value %.1f mfu %.4f gemm frac %.4f loss %.5f' % ('$1', d['ms_per_step'], d['value'], d['mfu'], d['roofline']['frac'], d['loss']))" || tail -3 gpurun_out/r3c35_$1.err; }
run c2_off "--stream-wt off" 10
run c2_on "--stream-wt on" 10
run c2_off_b "--stream-wt off" 10
run c2_on_b "--stream-wt on" 10
run l70 "--workload l70 --gemm-table gpurun_out/r3c35_l70_gemm_table.txt" 4
rocm-smi --showmeminfo vram 2>/dev/null | head -8
