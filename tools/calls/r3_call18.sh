#!/bin/bash
# round 3, call 18: which neighbour slows the forward GEMMs?  per-shape tables with single kernel classes skipped
mkdir -p gpurun_out
for m in 0 256 16 0 256; do
  python bench.py --steps 4 --warmup 3 --no-cpu-baseline --probe-skip $m --gemm-table gpurun_out/r3c18_table_skip$m.txt > gpurun_out/r3c18_skip$m.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c18_skip$m.json"))
t={}
for l in open("gpurun_out/r3c18_table_skip$m.txt"):
    if l.startswith("#"): continue
    f=l.split(); t[(f[0],f[1],f[2])]=float(f[7])
print("skip %3d  step %.2f  qkv %.1f  gateup %.1f  oproj/dq %.1f  down %.1f  dgrad14336 %.1f" % ($m, d["ms_per_step"], t[("2528","6144","4096")], t[("2528","28672","4096")], t[("2528","4096","4096")], t[("2528","4096","14336")], t[("2528","14336","4096")]))
PY
done
