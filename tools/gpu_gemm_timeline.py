"""GPU probe: where do the cycles of one K-tile go in the eight-phase GEMM?  Runs the timeline build of the 256 x 256 kernel
(tile variant 27 = gemm_nt_bf16_ph8_kernel<256, 2, MODE 4>): every wave stamps s_memtime at the start of each phase's load
section, at the start of its MFMA section and at its end; the first four blocks dump the stamps instead of their tile.
Prints, per phase and per wave-row (row 1 runs one barrier behind row 0), the median cycles of
   load  = MFMA start - load start   (ds_reads + LDS-DMA issue + lgkmcnt(0) + barrier wait)
   mfma  = MFMA end - MFMA start     (16 MFMAs: 256 cycles at the peak issue rate)
   gap   = next load start - MFMA end (second barrier + stamp bookkeeping)
and the K-tile period, next to the un-instrumented kernel's period derived from its launch time.
(written at the end of round 1 after the GPU budget was spent: the kernel compiles, its ISA was inspected, it has not run yet)
usage: PYTHONPATH=. python tools/gpu_gemm_timeline.py [M N K]"""
import os
os.environ.setdefault("UVX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "libuvx_probes.so"))  # probe tile variants live in the probes build
import sys
import torch
from ultravox_amd import ops, _lib

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (2528, 28672, 4096)
assert K // 64 <= 64, "the stamp buffer holds 64 K-tiles"
dev = "cuda"
L = _lib.lib()
torch.manual_seed(0)
a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
pool = [(torch.randn(N, K, device=dev) * 0.5).bfloat16() for _ in range(4)]      # cold weights, as in the training step


def timed(variant, reps=8):
    L.uvx_gemm_force_variant(variant)
    ops.gemm(a, pool[0])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(reps):
        out = ops.gemm(a, pool[(i + 1) % len(pool)])
    ev[1].record()
    torch.cuda.synchronize()
    L.uvx_gemm_force_variant(-1)
    return out, ev[0].elapsed_time(ev[1]) / reps * 1e3


_, us_plain = timed(11)
out, us_probe = timed(27)
nk = K // 64
raw = out.view(torch.int32).reshape(-1)[:4 * 8 * 64 * 12].cpu().to(torch.int64) & 0xFFFFFFFF
st = raw.view(4, 8, 64, 4, 3)[:, :, :nk]                       # [block, wave, K-tile, phase, (load start, MFMA start, MFMA end)]
d = lambda x, y: (x - y) & 0xFFFFFFFF                          # 32-bit wrap-safe difference
load = d(st[..., 1], st[..., 0]).float()
mfma = d(st[..., 2], st[..., 1]).float()
flat = st.reshape(4, 8, nk * 4, 3)
gap = d(flat[:, :, 1:, 0], flat[:, :, :-1, 2]).float()        # to the next phase's load start
gap = torch.cat([gap, gap[:, :, -1:]], dim=2).reshape(4, 8, nk, 4)
period = d(st[:, :, 1:, 0, 0], st[:, :, :-1, 0, 0]).float()   # K-tile period
steady = slice(2, nk - 2)                                       # skip pipeline fill / drain
tiles = ((M + 255) // 256) * ((N + 255) // 256)
rounds = -(-tiles // 256)
print(f"{M}x{N}x{K}: plain kernel {us_plain:.1f} us = {2.0 * M * N * K / us_plain / 1e6:.0f} TF/s, {rounds} round(s) of tiles -> "
      f"{us_plain / rounds / nk * 1e3:.0f} ns per K-tile incl. fill and epilogue; timeline build {us_probe:.1f} us ({us_probe / us_plain:.2f}x)")
for row, waves in (("row 0 (waves 0-3)", slice(0, 4)), ("row 1 (waves 4-7)", slice(4, 8))):
    print(f"--- {row}: median cycles over blocks 0-3, steady-state K-tiles")
    for ph in range(4):
        f = lambda x: x[:, waves, steady, ph].median().item()
        print(f"  phase {ph + 1}: load {f(load):6.0f}   mfma {f(mfma):6.0f}   gap {f(gap):6.0f}")
    print(f"  K-tile period {period[:, waves, steady].median().item():6.0f} cycles (MFMA floor 4 x 256 = 1024 per wave, 2048 per SIMD)")
print("stamps are s_memtime ticks (shader clock); compare load vs the other row's mfma: the design wants load <= mfma")
