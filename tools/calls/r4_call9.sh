#!/bin/bash
# round 4, call 9: tail of a partly filled round as a second launch with smaller tiles (cold probe + in-situ A/B of the tail-split threshold, option 10);
# rocprofv3 kernel stats of the 70B decode loop (C4); full-depth parity file with the tower calibration
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c9; mkdir -p $O
timeout 500 python tools/gpu_gemm_tailsplit_probe.py > $O/tailsplit.txt 2>&1
grep -v amdgpu.ids $O/tailsplit.txt
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
timeout 200 $B --gemm-table $O/table_A1.txt > $O/bench_A1.json 2>/dev/null
timeout 200 $B --opt 10=92 --gemm-table $O/table_B.txt > $O/bench_B_opt10_92.json 2>/dev/null
timeout 200 $B --opt 10=100 --gemm-table $O/table_C.txt > $O/bench_C_opt10_100.json 2>/dev/null
timeout 200 $B --opt 10=100 --gemm-override 2528x2560x4096=0 --gemm-table $O/table_D.txt > $O/bench_D_opt10_100_tail0.json 2>/dev/null
timeout 200 $B --gemm-table $O/table_A2.txt > $O/bench_A2.json 2>/dev/null
for f in A1 B_opt10_92 C_opt10_100 D_opt10_100_tail0 A2; do python - <<PY
import json
r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f", "ms/step", round(r["ms_per_step"],2), "gemm ms", round(r["roofline"]["gemm_ms_per_step"],2), "TF/s", round(r["roofline"]["achieved"],1), "loss", r["loss"])
PY
done
grep -h " 28672 \| 14336 " $O/table_A1.txt $O/table_B.txt $O/table_C.txt $O/table_D.txt | grep "2528 *\(28672\|14336\) *4096"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_decode70 -o d70 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 16 meta-llama/Llama-3.3-70B-Instruct > $GRAFT_REPO_ROOT/$O/decode70.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v "^W2026\|amdgpu.ids" $O/decode70.txt | tail -5
f=$(find $O/prof_decode70 -name "*kernel_stats*.csv" | head -1); head -25 "$f" | cut -c1-200
find $O/prof_decode70 -name "*kernel_trace*" -delete
timeout 900 python -m pytest tests/test_c2_full_depth_gpu.py -q > $O/pytest_full_depth.txt 2>&1
tail -5 $O/pytest_full_depth.txt
