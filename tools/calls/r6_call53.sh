#!/bin/bash
# round 6, call 53: the recipe flavours on the round's final build (CE, encoder LoRA r = 8, KL, KL + LoRA: one box) and the rocprofv3 kernel stats of the KL + LoRA step
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6c53; mkdir -p $O
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 ms/step', round(r['ms_per_step'],2), 'value', round(r['value'],1), 'loss', round(r['loss'],4), 'mfu', round(r['mfu'],4))"; }
cd $GRAFT_REPO_ROOT
for f in "ce:" "lora8:--audio-lora-r 8" "kl:--loss kl" "kl_lora8:--loss kl --audio-lora-r 8"; do
  name=${f%%:*}; flags=${f#*:}
  timeout 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-live-traffic $flags 2>$O/$name.err | tail -1 | line $name | tee -a $O/flavours.txt
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o kl --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-live-traffic --loss kl --audio-lora-r 8 > $O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
