#!/bin/bash
# round 4, call 12 (call 11 re-run after the bit_cast fix in dot8; its timings were of a kernel that read a quarter of the weights): the row-streaming decode GEMV + the 1024-thread decode attention: GPU tests, C4 bench line (B = 1, 8; A/B against the
# MFMA-mapping GEMV via option 4 = 2), rocprofv3 kernel stats of the 70B decode loop
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c12; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "few_rows" > $O/pytest_gemv.txt 2>&1; tail -3 $O/pytest_gemv.txt
timeout 600 python -m pytest tests/test_generate_gpu.py tests/test_baseline_configs_gpu.py tests/test_gemma_gpu.py tests/test_qwen_gpu.py tests/test_gemma3_gpu.py -x -q > $O/pytest_generate.txt 2>&1; tail -3 $O/pytest_generate.txt
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 > $O/bench_c4_b1.json 2> $O/bench_c4_b1.err; cat $O/bench_c4_b1.json; tail -2 $O/bench_c4_b1.err
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --opt 4=2 > $O/bench_c4_b1_mfma_gemv.json 2>/dev/null; cat $O/bench_c4_b1_mfma_gemv.json
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --batch 8 > $O/bench_c4_b8.json 2>/dev/null; cat $O/bench_c4_b8.json
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --batch 8 --opt 4=2 > $O/bench_c4_b8_mfma_gemv.json 2>/dev/null; cat $O/bench_c4_b8_mfma_gemv.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_decode70 -o d70 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 16 meta-llama/Llama-3.3-70B-Instruct > $GRAFT_REPO_ROOT/$O/decode70.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/decode70.txt | tail -4
