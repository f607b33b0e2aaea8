#!/bin/bash
# PMC passes over the LLM attention backward kernels (attn_bwd_dq_k / attn_bwd_dkdv_k at the C2 shape)
R=$PWD; OUT=$R/gpurun_out/pmc_attn_bwd; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/sq -o p --output-format csv -- timeout 300 python $R/tools/gpu_attn_bwd_run.py > $OUT/sq.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $OUT/lds -o p --output-format csv -- timeout 300 python $R/tools/gpu_attn_bwd_run.py > $OUT/lds.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS -d $OUT/more -o p --output-format csv -- timeout 300 python $R/tools/gpu_attn_bwd_run.py > $OUT/more.log 2>&1
python - <<PY
import csv, collections, os
for grp in ("sq", "lds", "more"):
    f = "$OUT/%s/p_counter_collection.csv" % grp
    if not os.path.exists(f): print(grp, "missing"); continue
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        for kn in ("attn_bwd_dq", "attn_bwd_dkdv"):
            if kn in r["Kernel_Name"]:
                agg[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(grp, k[0], k[1], "%.4g" % (sum(v) / len(v)), len(v))
    os.remove(f)
PY
rm -f $OUT/*/*kernel_trace.csv
