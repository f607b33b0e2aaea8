#!/bin/bash
# rocprofv3 --kernel-trace --stats over a short bench.py run; leaves s_kernel_stats.csv under gpurun_out/stats2/
R=$PWD; OUT=$R/gpurun_out/stats2; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-250
ls $OUT
rm -f $OUT/*kernel_trace.csv
