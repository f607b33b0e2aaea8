"""Build libuvx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

`python -m ultravox_amd.build` or `ultravox_amd.build.build()`; objects are rebuilt only when a
source or header is newer.  hipcc cross-compiles without a GPU, so this runs in the CPU container.

`--probes` (build(probes=True)) additionally links libuvx_probes.so: the same objects with gemm.hip compiled under
-DUVX_PROBES, which adds the superseded GEMM kernel families and the probe builds of the eight-phase kernel (tile variants
1-10, 12-14, 20-30) that tools/gpu_gemm_*.py measure.  The product loads libuvx.so only; the tools select the probes library
with UVX_LIB=ultravox_amd/libuvx_probes.so.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "build"
LIB = PKG / "libuvx.so"
LIB_PROBES = PKG / "libuvx_probes.so"
PROBE_SOURCES = ("gemm.hip", "attention.hip", "api_core.hip")   # the sources that read UVX_PROBES
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]
# per-file extras.  attention: keep MFMA results in VGPRs (gfx950 has a unified 512-entry file): the softmax
# consumes every accumulator element on the VALU, and the default AGPR form costs a v_accvgpr_read/write per
# element plus ~60 registers of occupancy.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libuvx.so for gfx950)")


def _newest_header() -> float:
    hdrs = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list((PKG.parent / "include").glob("*.h"))
    return max(h.stat().st_mtime for h in hdrs)


def _compile(src: Path, hipcc: str, hdr_mtime: float, force: bool, probes: bool = False) -> Path:
    obj = OBJ / (src.stem + ("_probes.o" if probes else ".o"))
    if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_mtime):
        return obj
    cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src.name, []), *(["-DUVX_PROBES"] if probes else []), "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr[-4000:]}")
    return obj


def _link(hipcc: str, lib: Path, objs, force: bool, verbose: bool) -> None:
    if force or not lib.exists() or any(o.stat().st_mtime > lib.stat().st_mtime for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(lib), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"linked {lib} from {len(objs)} objects")


def build(force: bool = False, verbose: bool = False, probes: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    hdr = _newest_header()
    jobs = [(s, False) for s in srcs] + ([(s, True) for s in srcs if s.name in PROBE_SOURCES] if probes else [])
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        done = dict(zip(jobs, ex.map(lambda j: _compile(j[0], hipcc, hdr, force, j[1]), jobs)))
    _link(hipcc, LIB, [done[(s, False)] for s in srcs], force, verbose)
    if probes:
        _link(hipcc, LIB_PROBES, [done[(s, s.name in PROBE_SOURCES)] for s in srcs], force, verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, probes="--probes" in sys.argv)
