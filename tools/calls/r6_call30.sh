#!/bin/bash
# round 6, call 30: which tile the backward's M = 2400 shapes want in situ (the picker's cost model was fitted at M = 2528): per-shape overrides of the C2 step
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r6c30; mkdir -p $O
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'gemm_ms', round(r['roofline']['gemm_ms_per_step'],2), 'loss', round(r['loss'],4))"; }
python - <<PY | tee $O/picks.txt
import ctypes
from ultravox_amd import _lib
l = _lib.lib()
for shp in ((2400, 14336, 4096), (2400, 4096, 28672), (2400, 4096, 4096), (2400, 4096, 6144)):
    print(shp, "picked variant", l.uvx_gemm_pick_variant(*shp))
PY
for rep in 1 2; do
for f in "default:" "down_dgrad_160:--gemm-override 2400x14336x4096=33" "down_dgrad_256:--gemm-override 2400x14336x4096=31" "down_dgrad_192:--gemm-override 2400x14336x4096=32" "gu_dgrad_192:--gemm-override 2400x4096x28672=32" "gu_dgrad_256:--gemm-override 2400x4096x28672=31"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>$O/$name.err | tail -1 | line $name | tee -a $O/m2400_tiles_ab.txt
done
done
