#!/bin/bash
# round 6, call 3: the merged-phase GEMM on 32 x 32 x 16 MFMAs (variants 61 / 62) - tests, the cold-weight probe against the 16 x 16 x 32 twins on the
# seven LLM + four encoder shapes, the sustained clock of both bodies (GRBM_GUI_ACTIVE / duration, separate --pmc pass), one in-situ arm (the q|k|v
# projection is the C2 launch that runs the 256-row tile with a plain epilogue); and persistent (35..38) vs plain merged-phase tiles on the encoder's
# K = 1024 shapes (libuvx_probes.so)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c3; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "mfma_32x32 or gelu or attention" 2>&1 | tail -5 | tee $O/pytest.txt
timeout 900 python tools/gpu_gemm_mfma32_probe.py 31,61,34,62,33 3 2>&1 | grep -v amdgpu.ids | tee $O/mfma32_probe.txt
# clocks: one --pmc pass over a short probe (31 vs 61 only, LLM shapes)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$O/pmc -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/gpu_gemm_mfma32_probe.py 31,61 1 > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY | tee $O/mfma32_clock.txt
import csv, glob, collections
cc = glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True)
kt = glob.glob("$O/pmc/**/*kernel_trace.csv", recursive=True)
if cc and kt:
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "gemm_nt_bf16_ph8" not in r["Kernel_Name"]:
            continue
        name, ns = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
        key = ("M32 " if "true>" in name.replace(" ", "") and name.count("true") else "") + name.split("(")[0][-60:] + f" grid={r['Grid_Size']}"
        a = agg[key]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += ns
    print("# GRBM_GUI_ACTIVE (summed over the 8 XCDs' GRBMs? see per-launch value / duration) per gemm launch class")
    for k, (n, c, ns) in sorted(agg.items()):
        print(f"{k:100s} launches {n:4d}  cycles/launch {c / n:12.0f}  us/launch {ns / n / 1e3:8.1f}  cycles/us {c / max(ns, 1) * 1e3:8.1f}")
else:
    print("no counter output", cc, kt)
PY
rm -rf $O/pmc
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'gemm_ms', round(r['roofline']['gemm_ms_per_step'],2))"; }
for rep in 1 2; do
for f in "ce_base:" "ce_qkv_m32:--gemm-override 2528x6144x4096=61"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>/dev/null | tail -1 | line $name | tee -a $O/insitu.txt
done
done
UVX_LIB=$GRAFT_REPO_ROOT/ultravox_amd/libuvx_probes.so timeout 600 python tools/gpu_gemm_cold_probe.py 31,32,33,34,35,36,37,38 enc 2>&1 | grep -v amdgpu.ids | tee $O/persistent_enc_probe.txt
