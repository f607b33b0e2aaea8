#!/bin/bash
# round 5, call 9: new tests (merged-tower export, generate under an un-merged LLM adapter, prefill rope + cache append in one launch);
# where the wav2vec2 tower's bf16 distance comes from (stage probe); which memory-side counters this rocprofv3 lists; c4s prefill with the
# fused rope + append against the pair (option 16); kernel stats of the 70B decode step at B = 8 and B = 32
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c9; mkdir -p $O
timeout 600 python -m pytest tests/test_lora_gpu.py tests/test_checkpoint_gpu.py tests/test_generate_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
timeout 300 python tools/gpu_c5_tower_stage_probe.py 1,4,12,24 10 > $O/c5_tower_stage_probe.txt 2>&1; grep -v amdgpu.ids $O/c5_tower_stage_probe.txt | tail -8
(rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) > $O/counters_all.txt 2>&1; grep -i -o "[A-Za-z0-9_]*\(DRAM\|MALL\|HBM\|UMC\|EA0_RD\|EA0_WR\|EA_RD\|EA_WR\)[A-Za-z0-9_]*" $O/counters_all.txt | sort -u > $O/counters_memory_side.txt; wc -l $O/counters_all.txt; cat $O/counters_memory_side.txt | tr '\n' ' '; echo; rm -f $O/counters_all.txt
for o in 1 0 1 0; do
  timeout 300 python bench.py --workload c4s --steps 4 --warmup 2 --opt 16=$o 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4s option16=$o prefill_ms', round(r['prefill_ms'],3), 'decode ms/token', round(r['decode_ms_per_token'],3))" | tee -a $O/c4s_rope_append_ab.txt
done
cd /tmp
for b in 8 32; do
  timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_dec$b -o d -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py $b 9 meta-llama/Llama-3.3-70B-Instruct > $GRAFT_REPO_ROOT/$O/decode70_b$b.txt 2>&1
  grep -v "^W2026\|amdgpu.ids" $GRAFT_REPO_ROOT/$O/decode70_b$b.txt | tail -2
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $GRAFT_REPO_ROOT/$O/prof_dec$b/d_results.db im2col_conv1_k > $GRAFT_REPO_ROOT/$O/decode70_b${b}_kernel_stats.txt 2>&1 || find $GRAFT_REPO_ROOT/$O/prof_dec$b | head
  head -24 $GRAFT_REPO_ROOT/$O/decode70_b${b}_kernel_stats.txt
  rm -rf $GRAFT_REPO_ROOT/$O/prof_dec$b
done
