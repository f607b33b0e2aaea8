#!/bin/bash
# round 4, call 13: decode attention reworked (key residue classes, 16-byte V loads) - whole GPU suite, C4 bench lines (B = 1, 8), C4s, kernel stats of the decode loop
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c13; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 > $O/bench_c4_b1.json 2> $O/bench_c4_b1.err; cat $O/bench_c4_b1.json | cut -c1-300
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --batch 8 > $O/bench_c4_b8.json 2>/dev/null; cat $O/bench_c4_b8.json | cut -c1-300
timeout 300 python bench.py --workload c4s --steps 3 --warmup 1 > $O/bench_c4s_b1.json 2>/dev/null; cat $O/bench_c4s_b1.json | cut -c1-300
timeout 300 python bench.py --workload c4s --steps 3 --warmup 1 --batch 8 > $O/bench_c4s_b8.json 2>/dev/null; cat $O/bench_c4s_b8.json | cut -c1-300
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_decode70 -o d70 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 16 meta-llama/Llama-3.3-70B-Instruct > $GRAFT_REPO_ROOT/$O/decode70.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/decode70.txt | tail -2
