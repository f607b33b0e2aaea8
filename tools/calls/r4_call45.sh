# round 4, call 45: bench.py with its new defaults (10 timed steps, 5 warm-up) on C2, next to call 44's 5 + 2 lines.
export PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c45
timeout 70 python bench.py --no-cpu-baseline > gpurun_out/r4c45/c2_default.json 2> gpurun_out/r4c45/c2_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c45/c2_default.json").read().strip().split("\n")[-1])
print("c2 default", d["steps"], d["warmup"], round(d["ms_per_step"], 2), d["roofline"]["gemm_ms_per_step"], d["roofline"]["frac"])
PY
