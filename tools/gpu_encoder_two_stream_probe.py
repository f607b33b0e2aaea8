"""GPU probe (round 4): does the frozen encoder forward (whisper-medium, 8 x 30 s) run faster as TWO half-batch chains on two streams
than as one chain?  (Its K = 1024 GEMMs spend a third of their time in pipeline fill and epilogue; a second chain's kernels can fill
the CUs a partly filled round leaves idle.)  Also: the whole encoder concurrently with a stand-in LLM-sized GEMM stream."""
import ctypes as C
import time
import torch
from ultravox_amd import _lib
from ultravox_amd._lib import check, ptr, stream_ptr
from ultravox_amd.config import UltravoxConfig
from ultravox_amd.model import UltravoxModel

dev = "cuda"
cfg = UltravoxConfig(audio_model_id="openai/whisper-medium", text_model_id="TinyLlama/TinyLlama-1.1B-Chat-v1.0", hidden_size=4096,
                     stack_factor=8, projector_ln_mid=True, torch_dtype="bfloat16")
model = UltravoxModel(cfg, device=dev, dtype=torch.bfloat16, seed=0, rope_len=512, with_backward=False)
l = _lib.lib()
F = 3000
mel = torch.randn(8, cfg.audio_config.num_mel_bins, F, device=dev).bfloat16()
lens = torch.full((8,), F, device=dev, dtype=torch.int64)


def enc(x, ln, out, ws, nb):
    check(l.uvx_encoder_fwd(stream_ptr(), C.byref(model._c), C.byref(model._ew), ptr(x), 0, ptr(ln), x.shape[0], F, ptr(out), ptr(ws),
                            C.c_size_t(nb)), "uvx_encoder_fwd")


def bufs(A):
    nb = l.uvx_encoder_ws_bytes(C.byref(model._c), A, F)
    return torch.empty(A, 1500, cfg.audio_config.d_model, device=dev, dtype=torch.bfloat16), torch.empty(nb, device=dev, dtype=torch.uint8), nb


o8, w8, n8 = bufs(8)
oa, wa, na = bufs(4)
ob, wb, nb_ = bufs(4)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    enc(mel, lens, o8, w8, n8)


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        enc(mel[:4], lens[:4], oa, wa, na)
    with torch.cuda.stream(s2):
        enc(mel[4:], lens[4:], ob, wb, nb_)
    cur.wait_stream(s1); cur.wait_stream(s2)


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(2):
    print(f"round {rnd}: one chain (8 clips) {timeit(one):.3f} ms   two chains (4 + 4 clips, two streams) {timeit(two):.3f} ms", flush=True)
one(); two(); torch.cuda.synchronize()
print("outputs agree:", torch.equal(o8[:4], oa), torch.equal(o8[4:], ob))
