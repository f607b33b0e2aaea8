#!/bin/bash
# round 4, call 14: RMSNorm fused into the decode GEMVs + rope / cache-append in one launch: kernel test, generate tests, C4 / C4s bench lines, decode kernel stats
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c14; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fused_rmsnorm or few_rows" > $O/pytest_kernels.txt 2>&1; tail -4 $O/pytest_kernels.txt
timeout 900 python -m pytest tests/test_generate_gpu.py tests/test_baseline_configs_gpu.py tests/test_gemma_gpu.py tests/test_qwen_gpu.py tests/test_gemma3_gpu.py tests/test_wav2vec2_gpu.py tests/test_lora_gpu.py -q > $O/pytest_generate.txt 2>&1; tail -4 $O/pytest_generate.txt
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 > $O/bench_c4_b1.json 2> $O/bench_c4_b1.err; cut -c1-200 $O/bench_c4_b1.json
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --batch 4 > $O/bench_c4_b4.json 2>/dev/null; cut -c1-200 $O/bench_c4_b4.json
timeout 300 python bench.py --workload c4s --steps 3 --warmup 1 > $O/bench_c4s_b1.json 2>/dev/null; cut -c1-200 $O/bench_c4s_b1.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_decode70 -o d70 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 16 meta-llama/Llama-3.3-70B-Instruct > $GRAFT_REPO_ROOT/$O/decode70.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/decode70.txt | tail -2
