#!/bin/bash
# round 5, call 28: do co-resident blocks help the encoder's K = 1024 GEMMs?  The legacy 128 x 128 four-wave kernel (variant 0: three blocks per CU, a
# weaker main loop) against the picked merged-phase tiles, in situ, per shape (bench.py --gemm-override / --gemm-table)
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c28; mkdir -p $O
timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --gemm-table $O/table_default.txt 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('default ms/step', round(r['ms_per_step'],2))"
timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --gemm-override 12000x4096x1024=0,12000x1024x1024=0,12000x3072x1024=0,12000x1024x4096=0 --gemm-table $O/table_v0.txt 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('encoder on variant 0 ms/step', round(r['ms_per_step'],2))"
timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --gemm-override 12000x4096x1024=34,12000x1024x1024=34,12000x3072x1024=34,12000x1024x4096=34 --gemm-table $O/table_v34.txt 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('encoder on variant 34 (128 x 256) ms/step', round(r['ms_per_step'],2))"
for t in default v0 v34; do echo "== $t"; grep "^ *12000" $O/table_$t.txt; done
