#!/bin/bash
# rocprofv3 --kernel-trace --stats over generate() at the C2 model size (tools/gpu_decode_probe.py)
R=$PWD; OUT=$R/gpurun_out/stats_decode; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 420 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- timeout 300 python $R/tools/gpu_decode_probe.py ${1:-8} 64 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-200
rm -f $OUT/*kernel_trace.csv
