"""GPU probe (round 6): the NN form of the merged-phase GEMM (B stored [K, N]: the dgrad on the forward weight as it lies, uvx_gemm_desc_t.b_kn) against the NT
kernel on the transposed copy, on the C2 step's four dgrad shapes and the 70B ones, COLD weights (a > 1 GB pool per shape), best of 3 rounds, one box."""
import torch
from ultravox_amd import ops, _lib

torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
# (M, N = N_in, K = N_out) of d x = d y . W, W [N_out, N_in]
shapes = [(2528, 4096, 6144), (2528, 4096, 4096), (2528, 4096, 28672), (2528, 14336, 4096), (632, 8192, 10240), (632, 8192, 57344), (632, 28672, 8192)]
for (M, N, K) in shapes:
    npool = min(48, max(2, -(-(1200 << 20) // (N * K * 2))))
    ws = [torch.randn(K, N, device=dev).bfloat16() for _ in range(npool)]        # forward weights [N_out, N_in] = [K, N]
    wts = [w.t().contiguous() for w in ws]
    a = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def run(fn):
        for i in range(npool):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(2):
            for i in range(npool):
                fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * npool) * 1e3

    rec = {}
    for rnd in range(3):
        rec["nt"] = min(rec.get("nt", 1e30), run(lambda i: ops.gemm(a, wts[i], out=out)))
        rec["nn"] = min(rec.get("nn", 1e30), run(lambda i: ops.gemm(a, ws[i], out=out, b_kn=True)))
    same = torch.equal(ops.gemm(a, ws[0], b_kn=True), ops.gemm(a, wts[0]))
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:7d} {K:7d} pool={npool:3d} | NT {rec['nt']:7.1f} us ({fl / rec['nt'] / 1e6:6.0f} TF/s)  NN {rec['nn']:7.1f} us ({fl / rec['nn'] / 1e6:6.0f} TF/s)  "
          f"NN/NT {rec['nn'] / rec['nt']:.3f}  identical={same}", flush=True)
    del ws, wts
