#!/bin/bash
# round 3, call 45: in-situ tile-variant check on the two C2 shapes whose tile count leaves a partial round (N = 28672: 1120 tiles of 256 x 256 = 4.4 rounds;
# N = 14336: 560 = 2.2 rounds): 160- and 192-row tiles fill rounds differently
mkdir -p gpurun_out/r3c45
sweep() {
  for v in default 31 32 33; do
    o=""; [ $v != default ] && o="--gemm-override $1=$v"
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --gemm-table gpurun_out/r3c45/t.txt $o > gpurun_out/r3c45/b.json 2>/dev/null
    python - "$1" "$v" <<'PY'
import sys, json
shape, v = sys.argv[1], sys.argv[2]
M, N, K = shape.split("x")
ms = json.load(open("gpurun_out/r3c45/b.json"))["ms_per_step"]
for line in open("gpurun_out/r3c45/t.txt"):
    f = line.split()
    if len(f) >= 9 and f[0] == M and f[1] == N and f[2] == K:
        print("%-18s variant %-8s picked %3s  avg_us %8.1f  TF/s %7.1f   step %.2f ms" % (shape, v, f[4], float(f[7]), float(f[8]), ms))
PY
  done
}
sweep 2528x28672x4096
sweep 2528x14336x4096
