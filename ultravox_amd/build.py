"""Build libuvx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

`python -m ultravox_amd.build` or `ultravox_amd.build.build()`; objects are rebuilt only when a
source or header is newer.  hipcc cross-compiles without a GPU, so this runs in the CPU container.
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
    hdrs = list(CSRC.glob("*.h")) + list((PKG.parent / "include").glob("*.h"))
    return max(h.stat().st_mtime for h in hdrs)


def _compile(src: Path, hipcc: str, hdr_mtime: float, force: bool) -> Path:
    obj = OBJ / (src.stem + ".o")
    if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_mtime):
        return obj
    cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr[-4000:]}")
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    hdr = _newest_header()
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(lambda s: _compile(s, hipcc, hdr, force), srcs))
    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"linked {LIB} from {len(objs)} objects")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
