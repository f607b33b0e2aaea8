#!/bin/bash
# round 5, call 2: the refitted split picker - parity tests, c4s / c4 bench lines at B = 1, 2, rocprofv3 kernel stats of the 8B and 70B
# prefill, FETCH_SIZE pass over the 70B prefill GEMMs (do the two row tiles of a weight panel share its fetch?)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_generate_gpu.py tests/test_baseline_configs_gpu.py -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python bench.py --workload c4s --steps 3 --warmup 2 > $O/bench_c4s_b1.json 2>$O/bench_c4s_b1.err
timeout 300 python bench.py --workload c4s --batch 2 --steps 3 --warmup 2 > $O/bench_c4s_b2.json 2>$O/bench_c4s_b2.err
timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 > $O/bench_c4_b1.json 2>$O/bench_c4_b1.err
timeout 600 python bench.py --workload c4 --batch 2 --steps 2 --warmup 1 > $O/bench_c4_b2.json 2>$O/bench_c4_b2.err
for f in c4s_b1 c4s_b2 c4_b1 c4_b2; do tail -1 $O/bench_$f.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$f prefill_ms', round(r['prefill_ms'],2), r['prefill']['tflops'], r['prefill']['gbps'], 'decode', round(r['decode_ms_per_token'],2))"; done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_prefill8 -o p8 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 2 > $GRAFT_REPO_ROOT/$O/prefill8.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_prefill70 -o p70 -- python $GRAFT_REPO_ROOT/tools/gpu_decode_probe.py 1 2 meta-llama/Llama-3.3-70B-Instruct > $GRAFT_REPO_ROOT/$O/prefill70.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/gpu_prefill_gemm_pmc.py 316 > $GRAFT_REPO_ROOT/$O/pmc_fetch.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v "^W2026\|amdgpu.ids" $O/prefill8.txt | tail -2; grep -v "^W2026\|amdgpu.ids" $O/prefill70.txt | tail -2
python tools/rocpd_stats.py $O/prof_prefill8/p8_results.db im2col_conv1_k 1 > $O/prefill8_kernel_stats.txt; head -16 $O/prefill8_kernel_stats.txt
python tools/rocpd_stats.py $O/prof_prefill70/p70_results.db im2col_conv1_k 1 > $O/prefill70_kernel_stats.txt; head -16 $O/prefill70_kernel_stats.txt
rm -f $O/prof_prefill8/*.db $O/prof_prefill70/*.db
python - <<PY
import csv, collections, glob
f = glob.glob("$O/pmc_fetch/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    gx = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0); wx = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)
    a = agg[(k, gx // max(wx, 1))]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open("$O/pmc_fetch_summary.txt", "w") as o:
    for (k, blocks), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        line = f"{k:50s} blocks {blocks:6d} launches {n:4d} FETCH_SIZE x2 per launch {2 * s / n / 1024:10.1f} MB"
        print(line); o.write(line + "\n")
PY
cat $O/pmc_fetch.txt | grep algorithmic
rm -rf $O/pmc_fetch
