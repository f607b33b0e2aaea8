#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/pmc_attn; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
python $R/tools/gpu_attn_pmc.py enc; python $R/tools/gpu_attn_pmc.py llm
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/sq -o p --output-format csv -- timeout 300 python $R/tools/gpu_attn_pmc.py enc > $OUT/sq.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/lds -o p --output-format csv -- timeout 300 python $R/tools/gpu_attn_pmc.py enc > $OUT/lds.log 2>&1
python - <<PY
import csv, collections
for grp in ("sq", "lds"):
    rows = list(csv.DictReader(open("$OUT/%s/p_counter_collection.csv" % grp)))
    agg = collections.defaultdict(list)
    for r in rows:
        if "attn_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(grp, k, "%.4g" % (sum(v) / len(v)), len(v))
PY
