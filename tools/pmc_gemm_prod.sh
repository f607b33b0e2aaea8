#!/bin/bash
# rocprofv3 PMC passes on the PRODUCTION GEMM (tile variant 31: merged-phase 256x256, static issue priority) and the kernel it
# replaced (variant 11: four phases per K-tile; VARS="11 30" PYVARS="11, 30" compares the issue-priority policies instead),
# shape 2528 x 28672 x 4096 (gate|up projection, C2).  Separate passes: the SQ block
# has 8 slots.  Counters only with --kernel-trace (no other trace domains).  Summary -> gpurun_out/pmc_prod/summary.txt
R=$PWD; OUT=$R/gpurun_out/pmc_prod; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
for V in ${VARS:-31 11}; do
  timeout 120 timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/sq_v$V -o p --output-format csv -- timeout 300 python $R/tools/gpu_gemm_pmc.py $V 2528 28672 4096 > $OUT/sq_v$V.log 2>&1
  timeout 120 timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/lds_v$V -o p --output-format csv -- timeout 300 python $R/tools/gpu_gemm_pmc.py $V 2528 28672 4096 > $OUT/lds_v$V.log 2>&1
done
python - <<PY > $OUT/summary.txt
import csv, collections, glob, os
out = "$OUT"
print("# rocprofv3 --pmc on gemm_nt_bf16_ph8_kernel<256,2,*> at 2528 x 28672 x 4096 bf16 (1120 tiles of 256x256, 64 K-tiles), 4 launches averaged")
print("# v31 = production (merged-phase: 2 sections of 32 MFMAs per K-tile); v11 = four-phase kernel (static issue priority); v30 = v11 with the round-1 priority policy")
print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES in cycles;")
print("# GRBM_GUI_ACTIVE summed over the 8 XCDs (per XCD / kernel duration = shader clock, here 1.9 GHz).  MFMA pipe utilisation is taken per RESIDENT wave pair: while a tile is resident 8 waves share 4 SIMDs.")
for v in (${PYVARS:-31, 11}):
    vals = collections.defaultdict(list); dur = []
    for d in ("sq", "lds"):
        for f in glob.glob(os.path.join(out, f"{d}_v{v}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_nt_bf16" in r["Kernel_Name"]:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(out, f"{d}_v{v}", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_nt_bf16" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"\n== variant {v}")
    for k in sorted(vals):
        print(f"  {k:28s} {sum(vals[k]) / len(vals[k]):14.4g}")
    if dur:
        print("  kernel durations under the profiler (us):", [round(x, 1) for x in dur])
    g = lambda k: sum(vals[k]) / len(vals[k]) if vals[k] else float("nan")
    wc = g("SQ_WAVE_CYCLES") * 4
    if wc == wc:
        # every wave needs one SIMD's MFMA pipe for its MFMAs; a CU runs 8 waves on 4 SIMDs, so pipe-cycles available = wave-cycles / 2
        print(f"  -> MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_WAVE_CYCLES x 4 / 2 waves per SIMD) = {g('SQ_VALU_MFMA_BUSY_CYCLES') / (wc / 2):.3f}")
        print(f"  -> wave time split: waiting (barrier / s_waitcnt) {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}, issue-stalled {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}, issuing {g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
        print(f"  -> LDS bank conflicts / LDS active cycles = {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.4f}")
PY
cat $OUT/summary.txt
rm -rf $OUT/sq_v* $OUT/lds_v*
