#!/bin/bash
# round 5, call 3: merged-phase kernels with three buffer sets (59 / 60) - bit-identity tests, cold-weight probe against their two-set twins
# on the prefill shapes; encoder attention at one / two clips with 64 vs 128 queries per block; bench lines
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r5c3; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "splitk or three_buffer or hand_scheduled" -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 500 python tools/gpu_gemm_splitk_probe.py all 316 33,59,34,60,18 > $O/splitk_probe_ns3.txt 2>&1; grep -v amdgpu.ids $O/splitk_probe_ns3.txt
timeout 300 python tools/gpu_gemm_splitk_probe.py 70b 632 33,59,34,60,31 > $O/splitk_probe_ns3_m632.txt 2>&1; grep -v amdgpu.ids $O/splitk_probe_ns3_m632.txt
python - > $O/enc_attn_qt.txt 2>&1 <<'PY'
import torch
from ultravox_amd import ops, _lib
L = _lib.lib(); dev = "cuda"; torch.manual_seed(0)
def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps * 1e3
for B in (1, 2, 4, 8):
    T, H, D = 1500, 16, 64
    qkv = (torch.randn(B, T, 3 * H * D, device=dev) * 0.5).bfloat16()
    q = qkv[..., :H * D].view(B, T, H, D); k = qkv[..., H * D:2 * H * D].view(B, T, H, D); v = qkv[..., 2 * H * D:].view(B, T, H, D)
    row = []
    for qt in (1, 2):
        L.uvx_attention_force_qt(qt)
        row.append(f"qt{qt}={timed(lambda: ops.attention(q, k, v, need_lse=False)):6.1f} us")
    L.uvx_attention_force_qt(0)
    print(f"encoder attention B={B}: " + " ".join(row), flush=True)
PY
cat $O/enc_attn_qt.txt | grep -v amdgpu
timeout 300 python bench.py --workload c4s --steps 3 --warmup 2 > $O/bench_c4s_b1.json 2>$O/bench_c4s_b1.err
timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 > $O/bench_c4_b1.json 2>$O/bench_c4_b1.err
for f in c4s_b1 c4_b1; do tail -1 $O/bench_$f.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$f prefill_ms', round(r['prefill_ms'],2), r['prefill']['tflops'], r['prefill']['gbps'], 'decode', round(r['decode_ms_per_token'],2))"; done
