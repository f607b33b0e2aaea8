#!/bin/bash
# round 4, call 16: LoRA small kernels (vectorised up-projection, register-resident down-projection, 4x the weight-gradient blocks): tests,
# the step flavours on one box, rocprofv3 kernel stats of the recipe flavour
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r4c16; mkdir -p $O
timeout 900 python -m pytest tests/test_lora_gpu.py tests/test_kl_gpu.py tests/test_kernels_gpu.py -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
timeout 200 $B > $O/bench_ce.json 2>/dev/null
timeout 200 $B --audio-lora-r 8 > $O/bench_lora8.json 2>/dev/null
timeout 200 $B --loss kl > $O/bench_kl.json 2>/dev/null
timeout 200 $B --loss kl --audio-lora-r 8 > $O/bench_kl_lora8.json 2>/dev/null
timeout 200 $B > $O/bench_ce_b.json 2>/dev/null
for f in ce lora8 kl kl_lora8 ce_b; do python - <<PY
import json
r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f", "ms/step", round(r["ms_per_step"],2), "value", round(r["value"],1), "mfu", round(r["mfu"],4), "loss", round(r["loss"],4))
PY
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_kl_lora8 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --loss kl --audio-lora-r 8 > $GRAFT_REPO_ROOT/$O/prof_run.txt 2>&1
cd $GRAFT_REPO_ROOT
