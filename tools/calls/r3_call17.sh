#!/bin/bash
# round 3, call 17: why are the K = 4096 forward GEMMs 20-30 % slower in the step than in any probe loop?  Per-launch times in issue order,
# with and without the non-GEMM kernels around them
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 3 --no-cpu-baseline --gemm-raw gpurun_out/r3c17_raw_base.txt --gemm-table gpurun_out/r3c17_table_base.txt > gpurun_out/r3c17_base.json 2>/dev/null
python bench.py --steps 4 --warmup 3 --no-cpu-baseline --probe-skip 255 --gemm-raw gpurun_out/r3c17_raw_skip.txt --gemm-table gpurun_out/r3c17_table_skip.txt > gpurun_out/r3c17_skip.json 2>/dev/null
head -12 gpurun_out/r3c17_table_base.txt; head -12 gpurun_out/r3c17_table_skip.txt
python - <<'PY'
for name in ("base", "skip"):
    rows = [l.split() for l in open(f"gpurun_out/r3c17_raw_{name}.txt")]
    per = len(rows) // 4
    step = rows[per:2 * per]
    sel = [float(r[5]) for r in step if r[:3] == ["2528", "6144", "4096"]]
    print(name, "2528x6144x4096 per layer (us):", " ".join(f"{x:.0f}" for x in sel))
    sel = [float(r[5]) for r in step if r[:3] == ["2528", "14336", "4096"]]
    print(name, "2528x14336x4096 per layer (us):", " ".join(f"{x:.0f}" for x in sel))
PY
