#!/bin/bash
# round 5, call 17: round-end evidence on the final build - tools/final_profile.sh (whole GPU suite, smoke, C2 bench + GEMM table, two-chain and
# autotuned arms, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes), then the inference lines (c4s, c4 at B = 1, 2, 8) and the prefill kernel stats
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
bash tools/final_profile.sh
O=gpurun_out/r5c17; mkdir -p $O
timeout 300 python bench.py --workload c4s --steps 4 --warmup 2 > $O/bench_c4s_b1.json 2>$O/bench_c4s_b1.err
timeout 300 python bench.py --workload c4s --batch 2 --steps 4 --warmup 2 > $O/bench_c4s_b2.json 2>$O/bench_c4s_b2.err
for b in 1 2 8; do timeout 600 python bench.py --workload c4 --batch $b --steps 2 --warmup 1 > $O/bench_c4_b$b.json 2>$O/bench_c4_b$b.err; done
for f in c4s_b1 c4s_b2 c4_b1 c4_b2 c4_b8; do tail -1 $O/bench_$f.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$f prefill_ms', round(r['prefill_ms'],2), 'tflops', round(r['prefill']['tflops'],1), 'decode ms/token', round(r['decode_ms_per_token'],2), 'frac', round(r['roofline']['frac'],4))"; done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_prefill8 -o p8 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 2 > $GRAFT_REPO_ROOT/$O/prefill8.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_prefill70 -o p70 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 2 meta-llama/Llama-3.3-70B-Instruct > $GRAFT_REPO_ROOT/$O/prefill70.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $O/prof_prefill8/p8_results.db im2col_conv1_k 1 > $O/prefill8_kernel_stats.txt; head -12 $O/prefill8_kernel_stats.txt
python tools/rocpd_stats.py $O/prof_prefill70/p70_results.db im2col_conv1_k 1 > $O/prefill70_kernel_stats.txt; head -12 $O/prefill70_kernel_stats.txt
rm -rf $O/prof_prefill8 $O/prof_prefill70
