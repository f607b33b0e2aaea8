#!/bin/bash
# round 5, call 1: split-K for the prefill GEMMs - parity tests, cold-weight probe of every (tile, split), rocprofv3 kernel stats of the
# 8B prefill, bench lines for c4s and c4 (70B) at B = 1
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c1; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "splitk or hand_scheduled or gemm" -p no:cacheprovider > $O/pytest_gemm.txt 2>&1; tail -4 $O/pytest_gemm.txt
timeout 400 python -m pytest tests/test_generate_gpu.py -q -x -p no:cacheprovider > $O/pytest_generate.txt 2>&1; tail -4 $O/pytest_generate.txt
timeout 400 python tools/gpu_gemm_splitk_probe.py all 316,632 > $O/splitk_probe.txt 2>&1; grep -v amdgpu.ids $O/splitk_probe.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_prefill8 -o p8 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 2 > $GRAFT_REPO_ROOT/$O/prefill8.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v "^W2026\|amdgpu.ids" $O/prefill8.txt | tail -4
f=$(find $O/prof_prefill8 -name "*kernel_stats*.csv" | head -1); head -40 "$f" | cut -c1-220
find $O/prof_prefill8 -name "*kernel_trace*" -delete
timeout 300 python bench.py --workload c4s --steps 3 --warmup 2 > $O/bench_c4s_b1.json 2>$O/bench_c4s_b1.err; tail -1 $O/bench_c4s_b1.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4s prefill_ms', r['prefill_ms'], 'decode', r['decode_ms_per_token'])"
timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 > $O/bench_c4_b1.json 2>$O/bench_c4_b1.err; tail -1 $O/bench_c4_b1.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4 prefill_ms', r['prefill_ms'], 'decode', r['decode_ms_per_token'])"
