#!/bin/bash
# round 3, call 40: in-situ tile-variant sweep on the encoder's K = 1024 / N = 1024 GEMM shapes (C2 step): per-shape avg us from the step's own HIP events
mkdir -p gpurun_out/r3c40
sweep() {  # shape
  for v in default 0 31 32 33 34; do
    o=""; [ $v != default ] && o="--gemm-override $1=$v"
    timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --gemm-table gpurun_out/r3c40/t.txt $o > gpurun_out/r3c40/b.json 2>/dev/null
    python - "$1" "$v" <<'PY'
import sys, json
shape, v = sys.argv[1], sys.argv[2]
M, N, K = shape.split("x")
ms = json.load(open("gpurun_out/r3c40/b.json"))["ms_per_step"]
for line in open("gpurun_out/r3c40/t.txt"):
    f = line.split()
    if len(f) >= 9 and f[0] == M and f[1] == N and f[2] == K:
        print("%-18s variant %-8s picked %3s  avg_us %8.1f  TF/s %7.1f   step %.2f ms" % (shape, v, f[4], float(f[7]), float(f[8]), ms))
PY
  done
}
sweep 12000x1024x1024
sweep 12000x4096x1024
sweep 12000x3072x1024
sweep 12000x1024x4096
