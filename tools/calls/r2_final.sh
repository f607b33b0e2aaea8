#!/bin/bash
# Round 2 evidence run: tools/final_profile.sh (tests, smoke, bench + GEMM table, rocprofv3 stats, PMC traffic), then the C5 bench line.
R=$PWD; export PYTHONPATH=$R
bash tools/final_profile.sh
timeout 400 python bench.py --workload c5 --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/bench_c5.log 2>&1; tail -1 $R/gpurun_out/final/bench_c5.log | cut -c1-400
