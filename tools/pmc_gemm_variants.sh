#!/bin/bash
# rocprofv3 PMC passes on GEMM tile variants at one shape: shader clock (GRBM_GUI_ACTIVE), MFMA pipe busy, wave time split.
# usage: VARS="31 49 55" SHAPE="2560 4096 8192" OUT=gpurun_out/xyz bash tools/pmc_gemm_variants.sh
R=$PWD; OUT=$R/${OUT:-gpurun_out/pmc_variants}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
SHAPE=${SHAPE:-2560 4096 8192}
for V in ${VARS:-31 49 55}; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/sq_v$V -o p --output-format csv -- timeout 150 python $R/tools/gpu_gemm_pmc.py $V $SHAPE > $OUT/sq_v$V.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/lds_v$V -o p --output-format csv -- timeout 150 python $R/tools/gpu_gemm_pmc.py $V $SHAPE > $OUT/lds_v$V.log 2>&1
done
python - <<PY > $OUT/summary.txt
import csv, collections, glob, os
out = "$OUT"
print("# rocprofv3 --pmc, GEMM shape $SHAPE bf16, 4 launches averaged per variant; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves;")
print("# SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs (/ 8 / duration = shader clock)")
for v in "${VARS:-31 49 55}".split():
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
    g = lambda k: sum(vals[k]) / len(vals[k]) if vals[k] else float("nan")
    d = sum(dur) / max(len(dur), 1)
    clk = g("GRBM_GUI_ACTIVE") / 8 / d / 1e3 if d else float("nan")
    nm = g("SQ_INSTS_MFMA")
    print(f"v{v}: {d:8.1f} us  clock {clk:5.2f} GHz  MFMA busy cycles / (16 x MFMA instrs) = {g('SQ_VALU_MFMA_BUSY_CYCLES') / (16 * nm) if nm else float('nan'):5.2f}  "
          f"MFMA-busy share of SQ_BUSY x 4 SIMD: {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * g('SQ_BUSY_CYCLES')):5.3f}  wave split: wait {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f} stall {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f} issue {g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}  "
          f"LDS active {g('SQ_LDS_IDX_ACTIVE'):.3g} conflicts {g('SQ_LDS_BANK_CONFLICT'):.3g}  SQ_BUSY {g('SQ_BUSY_CYCLES'):.4g} MFMA_BUSY {g('SQ_VALU_MFMA_BUSY_CYCLES'):.4g} WAVE_CYC {g('SQ_WAVE_CYCLES'):.4g}")
PY
cat $OUT/summary.txt
rm -rf $OUT/sq_v* $OUT/lds_v*
